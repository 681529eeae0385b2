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
orig0 = tpp_group.specs
def orig(a):          # + entries that are not part of the bench group: the same reductions on a matrix whose columns are NOT a power of two apart
    from tpp_group import Tpp, UNARY, UNARY_FLAG, DT
    m, n = 4096, 8192
    extra = [(f"reduce_cols_f32_ld{ld}", (lambda ld=ld: Tpp(a, f"unary REDUCE_X_OP_ADD over columns f32 {m} x {n} ld {ld}", "unary", UNARY.REDUCE_X_OP_ADD, m, n, ld, m, DT.F32, DT.F32,
                                                         UNARY_FLAG.REDUCE_COLS, out_elems=m, alg_bytes=4.0 * m * n + 4.0 * m))) for ld in (4160, 4352)]
    extra += [(f"reduce_rows_f32_ld{ld}", (lambda ld=ld: Tpp(a, f"unary REDUCE_X_OP_ADD over rows f32 {m} x {n} ld {ld}", "unary", UNARY.REDUCE_X_OP_ADD, m, n, ld, n, DT.F32, DT.F32,
                                                          UNARY_FLAG.REDUCE_ROWS, out_elems=n, alg_bytes=4.0 * m * n + 4.0 * n))) for ld in (4160,)]
    return orig0(a) + extra
if only:
    tpp_group.specs = lambda a: [s for s in orig(a) if s[0] in only]
    tpp_group._equation_orig, tpp_group._packed_orig = tpp_group._equation, tpp_group._packed
if os.environ.get("HINT"):            # the workloads' streaming hint (bench.timed sets the thread's hint from the workload it times)
    _Tpp_init = tpp_group.Tpp.__init__
    def _init(self, *a, **k):
        _Tpp_init(self, *a, **k); self.hint = int(os.environ["HINT"])
    tpp_group.Tpp.__init__ = _init
res = tpp_group.run(api, dev, 20, 0.2, 1.0, False, bench.timed)
for k, v in res.items():
    if only and k not in only:
        continue
    print(json.dumps({"tag": os.environ.get("TAG", ""), "label": k, **{x: v.get(x) for x in ("kernel", "us_per_launch", "frac_hbm", "verified", "error")}}), flush=True)
