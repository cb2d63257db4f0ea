#!/bin/bash
# AddressSanitizer + UBSan over the CPU code (GPU ASan is not available on this pool): the oracle (C) under its own CPU tests, and
# the host-side C++ mirror (bgs_demo) up to the point where it needs a GPU.  usage: bash tools/sanitize_cpu.sh
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/bgs_sanitize; mkdir -p $OUT
gcc -O1 -g -std=c11 -fPIC -ffp-contract=off -fno-fast-math -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -shared -Iinclude \
    -o $OUT/libbgs_oracle_asan.so oracle/bgs_oracle.c oracle/subsense_oracle.c oracle/dp_oracle.c oracle/ingest_oracle.c -lm
ASAN=$(gcc -print-file-name=libasan.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 BGS_ORACLE_LIB=$OUT/libbgs_oracle_asan.so \
    python -m pytest tests/test_oracle_cpu.py -x -q -p no:cacheprovider -k "not live_reference"  # (those run the reference-built sdLaMa091 as a child: the preload would reach it, and it reads uninitialised Vt bytes by design)
echo "oracle under ASan+UBSan: OK"
# host-side C++ mirror: XML handling, FrameProcessor set-up, the no-GPU exception path (a box without a GPU gets ONE std::exception)
g++ -O1 -g -std=c++14 -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -o $OUT/bgs_demo_asan tracking_amd/host/bgs_demo.cpp tracking_amd/host/FrameProcessor.cpp \
    -Ltracking_amd/lib -lbgs_hip -Wl,-rpath,$PWD/tracking_amd/lib
W=$OUT/work; rm -rf $W; mkdir -p $W/config
python - <<P
import numpy as np
np.random.default_rng(1).integers(0, 256, (3, 40, 52, 3), dtype=np.uint8).tofile("$W/frames.raw")
P
printf '<?xml version="1.0"?>\n<opencv_storage>\n<tictoc>""</tictoc>\n<enablePreProcessor>1</enablePreProcessor>\n<enableMixtureOfGaussianV2BGS>1</enableMixtureOfGaussianV2BGS>\n</opencv_storage>\n' > $W/config/FrameProcessor.xml
set +e
( cd $W && ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0 UBSAN_OPTIONS=halt_on_error=1 $OUT/bgs_demo_asan frames.raw 40 52 3 out > demo.log 2>&1 ); rc=$?
set -e
tail -3 $W/demo.log
if grep -q "ERROR: AddressSanitizer\|runtime error" $W/demo.log; then echo "host mirror under ASan+UBSan: FAILED"; exit 1; fi
echo "host mirror under ASan+UBSan: OK (exit code $rc: 1 = no GPU here, 0 = ran on a GPU)"
