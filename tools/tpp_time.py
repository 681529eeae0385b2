"""times the `tpp` group of bench.py alone (tools/tpp_group.py; one JSON line per entry with $TAG; $ONLY = comma-separated labels): the A/B harness of the TPP kernels"""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import bench, tpp_group, workloads as wl
from libxsmm_amd import capi
api = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
only = set(x for x in os.environ.get("ONLY", "").split(",") if x)
orig = tpp_group.specs
if only:
    tpp_group.specs = lambda a: [s for s in orig(a) if s[0] in only]
    tpp_group._equation_orig, tpp_group._packed_orig = tpp_group._equation, tpp_group._packed
res = tpp_group.run(api, dev, 20, 0.2, 1.0, False, bench.timed)
for k, v in res.items():
    if only and k not in only:
        continue
    print(json.dumps({"tag": os.environ.get("TAG", ""), "label": k, **{x: v.get(x) for x in ("kernel", "us_per_launch", "frac_hbm", "verified", "error")}}), flush=True)
