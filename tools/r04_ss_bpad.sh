# SuBSENSE 8 x 1080p: phase B held to fewer workgroups per CU (unused dynamic LDS) so that the post-processing chain beside it gets through
set -o pipefail
mkdir -p gpurun_out/ss_bpad
O=gpurun_out/ss_bpad/out.txt
: > $O
for rep in 1 2; do
for pad in 0 9000 14000 22000 30000; do
  echo "== BGS_SS_B_LDS_PAD=$pad" >> $O
  BGS_SS_B_LDS_PAD=$pad timeout -k 10 200 python tools/r04_ss_token.py --groups 1 >> $O 2>&1 || { echo FAILED >> $O; tail -5 $O; exit 1; }
done
done
grep -v amdgpu.ids $O
