R=$GRAFT_REPO_ROOT
for rf in 8 12 16 24; do for v in 16 24; do
  echo "== REFILL=$rf IPASS_MIN=$v"
  BGS_SS_REFILL=$rf BGS_SS_IPASS_MIN=$v python3 $R/tools/bench_configs.py --only subsense8both 2>&1 | grep -h "SuBSENSE" | sed 's/.*streams: //'
done; done
