# with phase B held to 4 workgroups per CU (the default since): ranges / token / parts once more
set -o pipefail
mkdir -p gpurun_out/ss_token2
O=gpurun_out/ss_token2/out.txt
: > $O
run() { echo "== $*" >> $O; env "$@" timeout -k 10 200 python tools/r04_ss_token.py $ARGS >> $O 2>&1 || { echo "FAILED: $*" >> $O; tail -5 $O; exit 1; }; }
ARGS="--groups 1,2"
run BGS_SS_A_TOKEN=0
run BGS_SS_A_TOKEN=1
ARGS="--groups 1"
run BGS_SS_PARTS=2
grep -E "^==|young|aged" $O
