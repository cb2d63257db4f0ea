# placement experiment: the MOG2 model built from fixed-size physical chunks (hipMemCreate / hipMemMap) instead of one hipMalloc
set -o pipefail
mkdir -p gpurun_out/r03i
for rep in 1 2 3; do
for mb in 2 16 64 256 1024 4096; do
    BGS_MODEL_VMM_CHUNK_MB=$mb timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 5 --main-only --no-pmc --no-cpu-baseline --sustain 0 --settle 100 > gpurun_out/r03i/vmm_${mb}_$rep.json 2> gpurun_out/r03i/vmm_${mb}_$rep.err || true
    python - "gpurun_out/r03i/vmm_${mb}_$rep.json" "chunk_MB=$mb" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d['ms_per_step'], d['roofline']['kernel_avg_ms'])
except Exception as e: print(sys.argv[2], 'failed', e, open(sys.argv[1].replace('.json','.err')).read()[-300:])
P
done
done
