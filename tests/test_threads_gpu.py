"""Concurrent dispatch + invocation from many host threads, as the reference's tests/threadsafety.c does: handles are
plain function pointers that any thread may call; dispatch hits are lock-free, misses are serialised; per-thread
execution state (stream, error) must not leak between threads.  [ref: tests/threadsafety.c; SURVEY 8(b) threading]"""
import threading

import numpy as np
import pytest

from helpers import GemmCase, TOL_F32, normf_rel
from libxsmm_amd import capi
from libxsmm_amd.capi import DT

pytestmark = pytest.mark.gpu

SHAPES = [(23, 23, 23), (32, 32, 32), (16, 16, 16), (64, 64, 64), (17, 40, 9), (32, 32, 32), (48, 16, 32), (23, 23, 23)]


def test_concurrent_dispatch_and_calls_from_eight_threads():
    api = capi.load()
    nthreads, rounds = 8, 6
    errors, handles = [], [dict() for _ in range(nthreads)]

    def worker(tid):
        try:
            import torch
            torch.cuda.set_device(0)
            for r in range(rounds):
                m, n, k = SHAPES[(tid + r) % len(SHAPES)]
                case = GemmCase(m, n, k, br_type=capi.BR_STRIDE, br_count=2, batch=3, seed=100 * tid + r)
                got, _, h = case.run_gpu(batched=True)
                ref, _ = case.run_oracle()
                err = normf_rel(case.valid_region(ref), case.valid_region(got), DT.F32)
                if not err < TOL_F32:
                    errors.append(f"thread {tid} round {r}: {m}x{n}x{k} normf_rel={err}")
                handles[tid][(m, n, k)] = h
                if api.hip_get_last_error() != 0:
                    errors.append(f"thread {tid}: {api.hip_get_last_error_string()}")
        except Exception as e:    # noqa: BLE001 -- reported to the main thread
            errors.append(f"thread {tid}: {e!r}")

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, "\n".join(errors)
    # the same descriptor dispatched from different threads yields the same registered handle
    seen = {}
    for d in handles:
        for shape, h in d.items():
            assert seen.setdefault(shape, h) == h
