import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import bench
from libxsmm_amd import capi
api = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
for dtype, m, batch in (("f32", 32, 4096), ("bf16", 32, 4096), ("bf16", 64, 4096), ("f32", 16, 4096), ("f32", 64, 4096)):
    for nsets, label in ((1, "resident"), (0, "rotated")):
        for hint in (0, 1, 2):          # 0: the library decides (launch size, or the recent launches of the thread together: round 6)
            w = bench.Workload(api, dev, dtype, m, batch, nsets=nsets, hint=hint)
            for i in range(3): w.step(i)
            torch.cuda.synchronize()
            _, _, us = bench.timed(w, 20, 0.2)
            print(json.dumps({"dtype": dtype, "m": m, "batch": batch, "sets": label, "nsets": w.nsets, "hint": hint, "kernel": w.kernel(), "us": round(us, 3)}), flush=True)
            del w; torch.cuda.empty_cache()
