# SuBSENSE 8 x 1080p: phase B limited to 4 / 3 waves per SIMD by the register allocation (-DBGS_SS_B_WAVES, no LDS pad) against the LDS pad (default build)
set -o pipefail
mkdir -p gpurun_out/ss_bwaves
O=gpurun_out/ss_bwaves/out.txt
: > $O
for rep in 1 2; do
  echo "== default build, BGS_SS_B_LDS_PAD=22000 (4 workgroups per CU, 157 KB of the CU's LDS taken)" >> $O
  timeout -k 10 200 python tools/r04_ss_token.py --groups 1 >> $O 2>&1 || exit 1
  for w in 4 3; do
    echo "== -DBGS_SS_B_WAVES=$w, BGS_SS_B_LDS_PAD=0" >> $O
    BGS_LIB_PATH=$PWD/tracking_amd/lib/abw$w/libbgs_hip.so BGS_SS_B_LDS_PAD=0 timeout -k 10 200 python tools/r04_ss_token.py --groups 1 >> $O 2>&1 || exit 1
  done
done
grep -E "^==|young|aged" $O
