set -o pipefail
for i in 1 2 3; do
timeout -k 10 600 env BGS_DEBUG_PROBE=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --main-only > gpurun_out/bp$i.json 2> gpurun_out/bp$i.err
grep "placement probe" gpurun_out/bp$i.err
python - gpurun_out/bp$i.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['sustained']['frac'], d['placement_probe'])
P
done
