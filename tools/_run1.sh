timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_batch2d_gpu.py -m gpu -q -x -p no:cacheprovider -k "long_reduction or 1x1" 2>&1 | tail -8
timeout 300 python - <<'PY'
import sys, json, torch
sys.path.insert(0, "."); import bench
from libxsmm_amd import capi
api = capi.load(); dev = torch.device("cuda:0"); torch.cuda.set_device(0)
api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
for br in (4096, 65536):
    w = bench.Workload(api, dev, "f32", 32, 1, br=br)
    r = bench.entry(w, 20, 0.2); r["br"] = br; print(json.dumps(r))
PY
