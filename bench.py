#!/usr/bin/env python
"""bench.py -- the headline metric of BASELINE.json on MI355X.

Workload (BASELINE.json configs[1]): strided BRGEMM, fp32, m=n=k=32, batch=4096 -- 4096 independent
(A_i, B_i, C_i) problems (the data-parallel batch axis of north_star) launched through ONE
libxsmm_dispatch_brgemm(STRIDE) handle with libxsmm_hip_gemm_batch_strided; every problem reduces
`--br` consecutive (A, B) pairs (default 1).  One "step" = one such launch over one batch.

Honest-roofline details:
  * inputs are resident in HBM before the timed region; the timed loop ROTATES over enough distinct
    input sets (default: > 2x the 256 MiB Infinity Cache) that every step streams from HBM and not
    from L3.  The L3-resident rate (same set every step) is reported separately as `l3_resident`.
  * the K steps of the timed region are captured once into a hipGraph (K kernel nodes, one per step) and
    replayed between the two barrier+synchronize pairs: the GPU sees K back-to-back launches and the Python
    interpreter is out of the loop.
  * roofline.achieved = algorithmic bytes per launch (br*(m*k*sA + k*n*sB) + m*n*sC*(1+[beta=1])
    per problem, times the batch) / mean launch duration = HIP-event time of the timed region (events recorded
    on the launch stream) / K.  It therefore includes the inter-kernel boundary and is a lower bound of what
    rocprofv3's per-kernel duration gives.
  * cpu_baseline: the reference's own JIT kernel (oracle/_ref, kind "reference") or, if that library is not
    present, the C restatement (kind "port"), single thread, on a bounded sample of the same workload.

N > 1 (launched by torch.distributed.run): one process per GPU, every rank owns its own batch (weak scaling,
no data-path collective); time = max over ranks between two barriers; value = total flops / time.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from libxsmm_amd import capi  # noqa: E402
from libxsmm_amd.capi import DT, GEMM_FLAG  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_PEAK_TF = {"f32": 157.3, "bf16": 2500.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--m", type=int, default=32, help="m = n = k")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--br", type=int, default=1, help="batch-reduce count of every problem")
    ap.add_argument("--beta", type=int, default=0, choices=[0, 1])
    ap.add_argument("--fused", type=int, default=0, help="1: bf16 column-bias + ReLU epilogue (config #5)")
    ap.add_argument("--sets", type=int, default=0, help="distinct input sets to rotate over (0: auto, > 2x L3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--cpu-threads", type=int, default=-1, help="threads of the all-core CPU leg (-1: one per physical core, 0/1: skip)")
    return ap.parse_args()


class Workload:
    def __init__(self, args, dev):
        self.args = args
        m, br, batch = args.m, args.br, args.batch
        self.bf16 = args.dtype == "bf16"
        es = 2 if self.bf16 else 4
        self.es = es
        tdt = torch.int16 if self.bf16 else torch.float32
        self.a_bytes = m * m * es
        self.flops_per_step = 2.0 * m * m * m * br * batch
        self.alg_bytes_per_step = float(batch) * (br * 2 * self.a_bytes + m * m * es * (1 + args.beta)) + (m * es if args.fused else 0)
        set_bytes = batch * (2 * br + 1) * self.a_bytes
        self.nsets = args.sets if args.sets > 0 else max(2, int(np.ceil(2.2 * 256 * 2 ** 20 / set_bytes)))
        g = torch.Generator(device="cpu").manual_seed(555)

        def values(n):   # reference-style data: multiples of 0.1 in [-0.4, 0.5]; bf16 by truncation
            v = torch.randint(-4, 6, (n,), generator=g).float() / 10
            if self.bf16:
                return (v.view(torch.int32) >> 16).to(torch.int16)
            return v
        n_ab = batch * br * m * m
        self.A = [values(n_ab).to(dev) for _ in range(self.nsets)]
        self.B = [values(n_ab).to(dev) for _ in range(self.nsets)]
        self.C = [torch.zeros(batch * m * m, dtype=tdt, device=dev) for _ in range(self.nsets)]
        self.D = values(m).to(dev) if args.fused else None
        api = capi.load()
        self.api = api
        t = DT.BF16 if self.bf16 else DT.F32
        flags = (0 if args.beta else GEMM_FLAG.BETA_0) | (GEMM_FLAG.VNNI_A if self.bf16 else 0)
        shape = capi.gemm_shape(m, m, m, m, m, m, t, t, t, DT.F32)
        cfg = capi.br_config(capi.BR_STRIDE, self.a_bytes, self.a_bytes, 0)
        self.shape, self.flags, self.cfg = shape, flags, cfg
        if args.fused:
            self.handle = api.dispatch_brgemm_ext(shape, flags, 0, cfg, capi.argops_cp(m, capi.UNARY.RELU, 0), capi.postops_colbias(m, t))
        else:
            self.handle = api.dispatch_brgemm(shape, flags, 0, cfg)
        if not self.handle:
            raise RuntimeError("dispatch returned NULL")
        self.brc = C.c_ulonglong(br)
        self.params = []
        for s in range(self.nsets):
            p = capi.GemmExtParam() if args.fused else capi.GemmParam()
            p.a.primary, p.b.primary, p.c.primary = self.A[s].data_ptr(), self.B[s].data_ptr(), self.C[s].data_ptr()
            p.op.tertiary = C.addressof(self.brc)
            if args.fused:
                p.d.primary = self.D.data_ptr()
            self.params.append(p)
        self.sa = br * self.a_bytes
        self.sc = m * m * es
        self.kernel = api.hip_kernel_name(self.handle, 1).decode()

    def step(self, s):
        p = self.params[s % self.nsets]
        if self.args.fused:
            self.api.hip_gemm_ext_batch_strided(self.handle, C.byref(p), self.args.batch, self.sa, self.sa, self.sc, 0, 0)
        else:
            self.api.hip_gemm_batch_strided(self.handle, C.byref(p), self.args.batch, self.sa, self.sa, self.sc)


def capture(work, steps, rotate):
    """Capture `steps` launches (one per step) into a hipGraph on a side stream; the timed region replays it,
    so the measurement sees K back-to-back kernel launches instead of K trips through the Python interpreter."""
    api = work.api
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        api.hip_set_stream(side.cuda_stream)
        for i in range(3):                      # warm the capture stream
            work.step(i if rotate else 0)
        side.synchronize()
        g.capture_begin()
        for i in range(steps):
            work.step(i if rotate else 0)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    return g


def timed(work, steps, barrier, rotate=True):
    """Exactly `steps` steps between two (barrier + synchronize) pairs.  Returns (wall seconds, mean
    microseconds per launch from HIP events recorded on the launch stream around the region).
    The steps are replayed from hipGraphs of at most 500 launches (a full chunk graph + a remainder graph)."""
    chunk = min(steps, 500)
    full, rem = divmod(steps, chunk)
    g_full = capture(work, chunk, rotate)
    g_rem = capture(work, rem, rotate) if rem else None
    g_full.replay()                               # untimed: first replay uploads the graph
    if g_rem is not None:
        g_rem.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(full):
        g_full.replay()
    if g_rem is not None:
        g_rem.replay()
    e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()                      # this rank's K steps are complete here; MAX over ranks is taken by the caller
    barrier()
    return t1 - t0, e0.elapsed_time(e1) * 1e3 / steps


def eager_kernel_us(work, steps, rotate=True):
    """Secondary view: every launch bracketed by its own event pair, issued eagerly from Python."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i in range(steps):
        evs[i][0].record(); work.step(i if rotate else 0); evs[i][1].record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3


def usable_cpus():
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota (containers)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // pr))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(args, seconds):
    """Reference JIT (or C restatement) on this box's host cores, single thread, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyoracle
    m, br = args.m, args.br
    bf16 = args.dtype == "bf16"
    es = 2 if bf16 else 4
    t = DT.BF16 if bf16 else DT.F32
    batch = min(args.batch, 1024)                       # private operands, streamed like the GPU run
    rng = np.random.default_rng(555)
    raw = ((np.floor(rng.random(batch * br * m * m * 2) * 10) - 4) / 10).astype(np.float32)
    ab = (raw.view(np.uint32) >> 16).astype(np.uint16) if bf16 else raw
    A, B = ab[: batch * br * m * m].copy(), ab[batch * br * m * m:].copy()
    Cc = np.zeros(batch * m * m, dtype=np.uint16 if bf16 else np.float32)
    flags = (0 if args.beta else GEMM_FLAG.BETA_0) | (GEMM_FLAG.VNNI_A if bf16 else 0)
    shape = capi.gemm_shape(m, m, m, m, m, m, t, t, t, DT.F32)
    cfg = capi.br_config(capi.BR_STRIDE, m * m * es, m * m * es, 0)
    brc = C.c_ulonglong(br)
    p = capi.GemmParam()
    p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = A.ctypes.data, B.ctypes.data, Cc.ctypes.data, C.addressof(brc)
    flops = 2.0 * m * m * m * br * batch
    if pyoracle.have_reference():
        ref = pyoracle.reference()
        h = ref.dispatch_brgemm(shape, flags, 0, cfg)
        if h:
            sa, sc = br * m * m * es, m * m * es
            t1 = ref.lib.xref_time_gemm_batch(h, C.byref(p), batch, sa, sa, sc, 5)
            reps = max(5, int(seconds / max(t1 / 5, 1e-9)))
            dt = ref.lib.xref_time_gemm_batch(h, C.byref(p), batch, sa, sa, sc, reps)
            out = {"value": round(flops * reps / dt / 1e9, 2), "unit": "GFLOP/s", "cores": 1, "kind": "reference",
                   "sample": f"reference JIT ({ref.lib.xref_get_target_arch().decode()}) kernel, {batch} problems x {reps} reps, private operands, 1 thread, {dt:.1f} s"}
            # the same kernel on every physical core at once (one thread per core, private operands): the caller's OpenMP loop of the
            # reference [samples/xgemm/gemm_kernel.c:4063-4066]; ctypes releases the GIL for the duration of each timing call
            nthreads = getattr(args, "cpu_threads", 0)
            if nthreads > 1:
                import threading
                sets = []
                for _ in range(nthreads):
                    a_, b_, c_ = A.copy(), B.copy(), Cc.copy()
                    q = capi.GemmParam()
                    q.a.primary, q.b.primary, q.c.primary, q.op.tertiary = a_.ctypes.data, b_.ctypes.data, c_.ctypes.data, C.addressof(brc)
                    sets.append((a_, b_, c_, q))
                def run_all(r):
                    times = [0.0] * nthreads

                    def work(i):
                        times[i] = ref.lib.xref_time_gemm_batch(h, C.byref(sets[i][3]), batch, sa, sa, sc, r)
                    ths = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
                    t0 = time.perf_counter()
                    for th in ths:
                        th.start()
                    for th in ths:
                        th.join()
                    return max(times), time.perf_counter() - t0
                pilot = max(2, reps // 100)                       # bounded whatever the container's CPU quota turns out to be
                _, wall_p = run_all(pilot)
                reps_mt = max(pilot, min(reps, int(pilot * 4.0 / max(wall_p, 1e-6))))      # aim at ~4 s of wall clock
                slowest, wall = run_all(reps_mt)
                agg = flops * reps_mt * nthreads / wall / 1e9
                out["all_cores"] = {"value": round(agg, 1), "unit": "GFLOP/s", "cores": nthreads, "speedup_vs_1_thread": round(agg / max(out["value"], 1e-9), 1),
                                    "sample": f"{nthreads} threads (usable CPUs of this container) x {batch} private problems x {reps_mt} reps, wall {wall:.1f} s"}
            return out
    # port: the C restatement (scalar loops); a much smaller sample keeps it bounded
    from helpers import GemmCase
    case = GemmCase(m, m, m, a_type=t, c_type=t, flags=GEMM_FLAG.VNNI_A if bf16 else 0, br_type=capi.BR_STRIDE, br_count=br, batch=64, seed=1)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < min(seconds, 5.0):
        case.run_oracle(); n += 1
    dt = time.perf_counter() - t0
    return {"value": round(2.0 * m ** 3 * br * 64 * n / dt / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": f"C restatement (oracle/), 64 problems x {n} reps, 1 thread, {dt:.1f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    api = capi.load()
    api.hip_set_device(local)
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)      # stream-ordered launches on torch's stream
    work = Workload(args, dev)

    for i in range(args.warmup):
        work.step(i)
    torch.cuda.synchronize()
    api.check()
    elapsed, kernel_us = timed(work, args.steps, barrier, rotate=True)
    api.check()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # secondary measurement: the same set every step (Infinity-Cache resident), not the headline
    for i in range(min(args.warmup, 10)):
        work.step(0)
    l3_elapsed, l3_kernel_us = timed(work, args.steps, barrier, rotate=False)
    eager_us = eager_kernel_us(work, min(args.steps, 100))

    if rank == 0:
        # HBM traffic per launch from the TCC counters: rocprofv3 --pmc cannot run inside this process, so the number
        # comes from the committed PMC summary of the same workload (tools/profile_paths.sh, separate FETCH_SIZE /
        # WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 correction) -- null when no entry matches.
        traffic, traffic_src = None, None
        for f in sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            try:
                for w in json.load(open(f))["workloads"]:
                    if w["algorithmic_bytes_per_launch"] == int(work.alg_bytes_per_step) and w["kernel"].replace(" ", "") == api.hip_kernel_name(work.handle, 1).decode().replace(" ", ""):
                        traffic, traffic_src = w["traffic_bytes_per_launch"], os.path.relpath(f, ROOT)
                        break
            except Exception:
                continue
            if traffic is not None:
                break
        total_flops = work.flops_per_step * args.steps * world
        value = total_flops / elapsed / 1e9
        gbs = work.alg_bytes_per_step / (kernel_us * 1e-6) / 1e9
        peak_tf = MFMA_PEAK_TF[args.dtype]
        out = {
            "metric": f"GFLOP/s, batched stride-BRGEMM m=n=k={args.m} {args.dtype}",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"stride-BRGEMM {args.dtype} m=n=k={args.m}, batch={args.batch} independent problems per GPU, br={args.br}, beta={args.beta}"
                                   + (", fused colbias+ReLU" if args.fused else ""),
                       "kernel": work.api.hip_kernel_name(work.handle, 1).decode(), "input_sets_rotated": work.nsets, "per_gpu_batch": args.batch},
            "pct_mfma_peak": round(100.0 * value / world / 1e3 / peak_tf, 2),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_us": round(kernel_us, 3), "kernel_us_eager_event_pairs": round(eager_us, 3),
                         "algorithmic_bytes_per_launch": int(work.alg_bytes_per_step),
                         "note": "kernel_us = HIP-event time of the timed region / steps (includes the ~1.5 us inter-kernel boundary)"},
            "l3_resident": {"value": round(work.flops_per_step * args.steps * world / l3_elapsed / 1e9, 1), "unit": "GFLOP/s",
                            "kernel_us": round(l3_kernel_us, 3),
                            "achieved_GBs": round(work.alg_bytes_per_step / (l3_kernel_us * 1e-6) / 1e9, 1)},
        }
        if not args.no_cpu_baseline and world == 1:
            if args.cpu_threads < 0:
                args.cpu_threads = usable_cpus()
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
