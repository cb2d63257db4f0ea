set -o pipefail
mkdir -p gpurun_out/r03g
python - <<'P' > gpurun_out/r03g/configs.json
import json, sys
sys.path.insert(0, '.')
from tools import bench_configs
print(json.dumps(bench_configs.configs_block(cpu=False)))
P
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03g/configs.json').read().strip().splitlines()[-1])
c=d['configs2_wmv_abl_4k']
for k in ('wmv','abl','both_kernels','fused_group'): print(k, json.dumps(c[k]))
P
# PMC traffic of the fused launch
cat > /tmp/fan_run.py <<'P'
import sys, torch
sys.path.insert(0, sys.argv[1])
from tools import synth
from tracking_amd import capi
from tracking_amd.engine import Group
S, rows, cols, T = 8, 2160, 3840, 8
pool = torch.stack([synth.s_surv(T, rows, cols, seed=4321 + s, device="cuda") for s in range(S)], 1)
g = Group([capi.WMV, capi.ABL], n_streams=S); g.set_geometry(rows, cols, 3); g.set_option(capi.OPT_BORROW_FRAMES, 1)
fg1 = torch.empty((S, rows, cols), dtype=torch.uint8, device="cuda"); fg2 = torch.empty_like(fg1)
for t in range(30): g.process_batch_device(pool[t % T], [fg1, fg2], None)
torch.cuda.synchronize()
P
for c in FETCH_SIZE WRITE_SIZE; do bash tools/pmc_kernel.sh fan_$c fan_kernel "$c" -- /tmp/fan_run.py $GRAFT_REPO_ROOT; done
