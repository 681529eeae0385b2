"""scratch: f32 BCSC stream kernel vs the gold loop, error pattern by (M-block, column, row)"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG
from oracle import pyoracle
from test_sparse_gpu import make_bcsc
api, orc = capi.load(), pyoracle.oracle()
M, N, K, mb, bk, bn = 64, 64, 64, 8, 16, 16
rng = np.random.default_rng(1)
colptr, rowidx, bvals = make_bcsc(rng, K, N, bk, bn, 0.6, DT.F32)
print("colptr", colptr, "rowidx", rowidx)
# A[mb][k][i] = 1000 mb + k + i / 100 ; B = 1 -> C[mb][n][i] = sum over the column block's k-blocks
A = np.zeros((mb, K, M), dtype=np.float32)
for b in range(mb):
    for k in range(K):
        A[b, k, :] = (b + 1) * 1.0 + 0.0 * k
A = rng.integers(-3, 4, (mb, K, M)).astype(np.float32)
Bv = rng.integers(-3, 4, bvals.size).astype(np.float32)
ref = np.zeros(mb * N * M, dtype=np.float32)
orc.lib.oracle_packed_spgemm_bcsc(DT.F32, DT.F32, M, N, K, mb, bk, bn, 0, A.ctypes.data, Bv.ctypes.data, colptr.ctypes.data, rowidx.ctypes.data, ref.ctypes.data, 1)
h = api.create_packed_spgemm_bcsc(capi.gemm_shape(mb, 0, K, K, 0, N, DT.F32, DT.F32, DT.F32, DT.F32), GEMM_FLAG.BETA_0, 0, capi.SpgemmConfig(M, bk, bn))
dA, dB = torch.from_numpy(A.reshape(-1)).cuda(), torch.from_numpy(Bv).cuda()
dC = torch.zeros(mb * N * M, dtype=torch.float32, device="cuda")
dcp, dri = torch.from_numpy(colptr.view(np.int32)).cuda(), torch.from_numpy(rowidx.view(np.int32)).cuda()
nblk = C.c_ulonglong(N // bn)
p = capi.GemmParam()
p.a.primary, p.b.primary, p.b.secondary, p.b.tertiary, p.b.quaternary, p.c.primary = dA.data_ptr(), dB.data_ptr(), dcp.data_ptr(), dri.data_ptr(), C.addressof(nblk), dC.data_ptr()
capi.Api.call(h, p); api.hip_sync(); api.check()
print(api.hip_kernel_name(h, 0))
got = dC.cpu().numpy().reshape(mb, N, M); ref = ref.reshape(mb, N, M)
bad = got != ref
print("bad fraction", bad.mean(), "per mb", bad.reshape(mb, -1).mean(1), "per col(n)", bad.transpose(1, 0, 2).reshape(N, -1).mean(1).round(2), "per row(i)", bad.transpose(2, 0, 1).reshape(M, -1).mean(1).round(2))
print("got[0,0,:8]", got[0, 0, :8], "ref", ref[0, 0, :8])
print("got[0,17,:8]", got[0, 17, :8], "ref", ref[0, 17, :8])
# is got a permutation / other block of ref?
for b in range(mb):
    for b2 in range(mb):
        if np.array_equal(got[b], ref[b2]) and b != b2:
            print("got block", b, "== ref block", b2)
