#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory into a small JSON + text table (the part committed under profiles/).

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are reported in
KiB-units of 1024 B by rocprofv3 (hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024), collected in separate --pmc passes, and
on gfx950 FETCH_SIZE counts exactly half of the bytes of a wide coalesced streaming read -> doubled.
usage: prof_summary.py <prof_dir> <kernel-substring> <min_grid> [algorithmic_bytes_per_launch] [last_n] [skip_last]
last_n / skip_last: drop the last skip_last matching dispatches of every run, then keep the last n.  bench.py --main-only ends
with K timed launches followed by --sustain further ones: (last_n, skip_last) = (K, sustain) is the timed region, (sustain, 0)
the sustained leg.
"""
import collections
import csv
import json
import os
import sys


def rows(path):
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


def grid(r):
    return int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))


def main():
    d, sub, min_grid = sys.argv[1], sys.argv[2], int(sys.argv[3])
    algo = float(sys.argv[4]) if len(sys.argv) > 4 else None
    last_n = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    skip = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    out = {"kernel": sub, "min_grid": min_grid, "last_n_dispatches": last_n, "after_dropping_last": skip}

    def window(v):
        v = v[:len(v) - skip] if skip else v
        return v[-last_n:] if last_n else v

    kt = [r for r in rows(os.path.join(d, "stats", "stats_kernel_trace.csv")) if sub in r["Kernel_Name"] and grid(r) >= min_grid]
    if kt:
        kt.sort(key=lambda r: int(r["Start_Timestamp"]))
        kt = window(kt)
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt]
        out.update(launches=len(dur), avg_us=sum(dur) / len(dur) / 1e3, min_us=min(dur) / 1e3, max_us=max(dur) / 1e3,
                   grid=grid(kt[0]), vgpr=int(kt[0]["VGPR_Count"]), sgpr=int(kt[0]["SGPR_Count"]), name=kt[0]["Kernel_Name"])
    ctr = {}
    for sub_dir in sorted(os.listdir(d)):
        p = os.path.join(d, sub_dir)
        if not (os.path.isdir(p) and sub_dir.startswith("pmc_")):
            continue
        for f in os.listdir(p):
            if f.endswith("counter_collection.csv"):
                agg = collections.defaultdict(list)
                rs = [r for r in rows(os.path.join(p, f)) if sub in r["Kernel_Name"] and grid(r) >= min_grid]
                rs.sort(key=lambda r: int(r["Dispatch_Id"]))
                for r in rs:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                for k, v in agg.items():
                    v = window(v)
                    ctr[k] = sum(v) / len(v)
    out["counters_avg_per_launch"] = ctr
    if "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
        rd, wr = ctr["FETCH_SIZE"] * 1024 * 2, ctr["WRITE_SIZE"] * 1024
        out["hbm_read_bytes_per_launch (FETCH_SIZE*1024*2, gfx950 correction)"] = rd
        out["hbm_write_bytes_per_launch (WRITE_SIZE*1024)"] = wr
        out["hbm_bytes_per_launch"] = rd + wr
        if algo:
            out["algorithmic_bytes_per_launch"] = algo
            out["traffic_over_algorithmic"] = (rd + wr) / algo
    if algo and "avg_us" in out:
        out["algorithmic_GBps"] = algo / (out["avg_us"] * 1e-6) / 1e9
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
