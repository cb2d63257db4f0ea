# SuBSENSE 8 x 1080p: the phase A token and the cut of one batch into parts, each setting in a process of its own (gpurun -- bash tools/r04_ss_token.sh)
set -o pipefail
mkdir -p gpurun_out/ss_token
O=gpurun_out/ss_token/out.txt
: > $O
run() { echo "== $*" >> $O; env "$@" timeout -k 10 200 python tools/r04_ss_token.py $ARGS >> $O 2>&1 || { echo "FAILED: $*" >> $O; tail -5 $O; exit 1; }; }
ARGS="--groups 1,2,4"
run BGS_SS_A_TOKEN=0
run BGS_SS_A_TOKEN=1
ARGS="--groups 1"
run BGS_SS_PARTS=2
run BGS_SS_PARTS=4
run BGS_SS_PARTS=4 GPU_MAX_HW_QUEUES=4
ARGS="--groups 4"
run BGS_SS_A_TOKEN=1 GPU_MAX_HW_QUEUES=4
cat $O
