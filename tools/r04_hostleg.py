"""Round 4 diagnostic: bench.py's host_path leg in a FRESH process, after nothing / after the copy calibration / after 20 GB of torch
allocations have come and gone - which part of the process's history makes 16 separately page-locked camera buffers slow?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tools import synth
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
clip = synth.s_sat(25, 1080, 1920, seed=1234, device="cuda").cpu().numpy()
if mode == "calib":
    print(bench.calibrate(0, 32))
if mode == "torch":
    x = [torch.empty(5 << 30, dtype=torch.uint8, device="cuda") for _ in range(4)]
    del x
    torch.cuda.empty_cache()
if mode == "engine":  # a big engine has lived and died in this process
    from tracking_amd import Engine, capi
    e = Engine(capi.MOG2, n_streams=32); e.set_geometry(1080, 1920, 3)
    fg = torch.empty((32, 1080, 1920), dtype=torch.uint8, device="cuda")
    fr = torch.zeros((32, 1080, 1920, 3), dtype=torch.uint8, device="cuda")
    for _ in range(5): e.process_batch_device(fr, fg, None, None)
    torch.cuda.synchronize(); e.close(); del fr, fg; torch.cuda.empty_cache()
h = bench.host_leg(0, clip)
print(mode, {k: (v["ms_per_frame"], v.get("bus_GBps")) for k, v in h.items() if isinstance(v, dict) and "ms_per_frame" in v})
