#!/bin/bash
# Per-dispatch counters of mog2_update_kernel over the timed (fresh frames) and sustain (the same 25 frames repeating) launches of
# bench.py --main-only: what changes between 1.40 ms on never-seen noise and 1.12 ms on the 8th repetition?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_pmc_series
mkdir -p $OUT
if [ -n "$PMC_SET" ]; then SETS=("$PMC_SET"); else SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"); fi
: > $OUT/summary.txt
for ctrs in "${SETS[@]}"; do
  tag=$(echo $ctrs | tr ' ' '_' | cut -c1-40)
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/raw_$tag -o pmc -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-pmc --no-cpu-baseline > $OUT/run_$tag.log 2>&1
  python3 - "$OUT/raw_$tag" "$ctrs" <<'PY' | tee -a $OUT/summary.txt
import csv, glob, sys, collections
out, ctrs = sys.argv[1], sys.argv[2].split()
rows = []
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "mog2_update" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)[-220:]   # 20 timed + 200 sustain
def mean(sel, c):
    v = [by[i].get(c, 0.0) for i in sel]
    return sum(v) / max(1, len(v))
print("== counters:", " ".join(ctrs), " (update-kernel dispatches: %d)" % len(by))
for c in sorted({c for i in ids for c in by[i]}):
    print("%-26s timed(fresh) %16.1f | sustain cycle1 %16.1f | cycle4 %16.1f | cycle8 %16.1f" % (c, mean(ids[:20], c), mean(ids[20:45], c), mean(ids[95:120], c), mean(ids[195:220], c)))
PY
  rm -rf $OUT/raw_$tag
done
