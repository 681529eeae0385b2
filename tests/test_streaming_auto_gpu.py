"""The automatic streaming decision (libxsmm_hip_set_streaming_hint(0), include/libxsmm_hip.h): a launch smaller than the Infinity Cache keeps cacheable operand
loads while the calling thread keeps launching on the same operand set, takes non-temporal loads once the thread's recent launches walk over more operand sets than
the 256 MiB cache holds, and goes back when the caller settles on one set again.  Either way the results are the same numbers (the policy is a cache hint)."""
import ctypes as C

import pytest

from libxsmm_amd import capi
from libxsmm_amd.capi import DT, GEMM_FLAG as F

pytestmark = pytest.mark.gpu


def test_walking_over_more_operand_sets_than_the_cache_holds_switches_to_streaming_and_back():
    import torch
    api = capi.load()
    batch, m = 4096, 32
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(9)
    nsets = 8                                                # 8 x 48 MiB = 384 MiB of operands: more than the cache
    As = [(torch.randint(-4, 6, (batch, m, m), generator=g).float() / 10).to(dev) for _ in range(nsets)]
    Bs = [torch.roll(a, 1, 0) for a in As]
    Cs = [torch.empty_like(a) for a in As]
    shape = capi.gemm_shape(m, m, m, m, m, m, DT.F32, DT.F32, DT.F32, DT.F32)
    h = api.dispatch_brgemm(shape, F.BETA_0, 0, capi.br_config(capi.BR_STRIDE, m * m * 4, m * m * 4, 0))
    assert h
    cnt = C.c_ulonglong(1)

    def launch(s):
        p = capi.GemmParam()
        p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = As[s].data_ptr(), Bs[s].data_ptr(), Cs[s].data_ptr(), C.addressof(cnt)
        api.hip_gemm_batch_strided(h, C.byref(p), batch, m * m * 4, m * m * 4, m * m * 4)
        api.check()
        return api.hip_streaming_window_verdict()
    api.hip_set_streaming_hint(0)
    for _ in range(200):                                     # whatever earlier tests of this process left in the window is forgotten
        v = launch(0)
    assert v == 0                                            # one resident set: cacheable
    api.hip_sync()
    cached = Cs[0].clone()
    verdicts = [launch(i % nsets) for i in range(3 * nsets)]
    assert verdicts[-1] == 1 and verdicts[0] == 0            # ... a walk over 384 MiB: streaming, from the sixth distinct set on
    assert verdicts.index(1) == 5
    api.hip_sync()
    assert torch.equal(Cs[0], cached)                        # the same numbers under either policy
    # a hand-over inside a chain: a TPP whose input is what the GEMM just wrote stays cacheable although the walk still fills the window
    from libxsmm_amd.capi import UNARY
    verdicts = [launch(i % nsets) for i in range(nsets)]
    assert verdicts[-1] == 1
    ht = api.dispatch_meltw_unary(UNARY.RELU, capi.UnaryShape(m * m, batch, m * m, m * m, DT.F32, DT.F32, DT.F32), 0)
    assert ht
    out = torch.empty_like(Cs[0])
    up = capi.UnaryParam(); up.in_.primary, up.out.primary = Cs[nsets - 1].data_ptr(), out.data_ptr()
    capi.Api.call(ht, up)
    api.check()
    assert api.hip_streaming_window_verdict() == 0
    up.in_.primary = As[2].data_ptr()                        # ... while the same TPP on an operand nobody just produced streams like the rest of the walk
    capi.Api.call(ht, up)
    api.check()
    assert api.hip_streaming_window_verdict() == 1
    api.hip_sync()
    assert torch.equal(out, torch.relu(As[2]))
    for _ in range(200):
        v = launch(3)
    assert v == 0                                            # settled on one set again
    api.hip_sync(); api.check()
