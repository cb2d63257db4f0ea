#!/bin/bash
# MOG1 on fresh-noise frames: this tree against tracking_amd/lib/abm1 (the tree before the per-wave rank count), alternating on one box
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  echo "== current"; python3 $R/tools/r04_mog1_fresh.py 2>&1 | grep MOG1
  echo "== before (lib/abm1)"; BGS_LIB_PATH=$R/tracking_amd/lib/abm1/libbgs_hip.so BGS_LIB_PARTIAL_ABI=1 python3 $R/tools/r04_mog1_fresh.py 2>&1 | grep MOG1
done
