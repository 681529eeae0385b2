#!/usr/bin/env python
"""bench.py -- the headline metric of BASELINE.json on MI355X, plus the sweep the metric names.

Headline (the `value`): BASELINE.json configs[1] -- strided BRGEMM, fp32, m=n=k=32, batch=4096 independent
(A_i, B_i, C_i) problems launched through ONE libxsmm_dispatch_brgemm(STRIDE) handle with
libxsmm_hip_gemm_batch_strided.  One "step" = one such launch over one batch.

Output contract: the LAST stdout line is one compact JSON object (< 3.5 KB: the driver keeps an 8 KB tail) with the headline, its roofline,
the CPU baseline and one or two numbers per secondary workload; the full record of the run (every kernel name, time, verification norm,
CPU sample description) goes to bench_detail.json next to this file and to stderr BEFORE that line.

What the record carries (all measured in this process, on this GPU):
  * value / ms_per_step: the timed region is a hipGraph of back-to-back launches (a multiple of --steps), replayed until
    it lasts >= --min-seconds (0.5 s): `steps` echoes the flag, `steps_timed` is what was really timed.  Inputs are
    resident in HBM and ROTATE over enough distinct sets (> 2x the 256 MiB Infinity Cache) that every step streams from HBM.
  * roofline: algorithmic bytes (br*(m*k*sA + k*n*sB) + m*n*sC*(1+[beta=1]) per problem) / HIP-event time per launch.
    `traffic` = TCC-counter bytes per launch from the committed PMC pass of this same command (profiles/), null if none.
  * verified: after the timed region a strided sample of C is compared with the oracle (oracle/liboracle.so) on the same inputs.
  * sweep: m in {16,32,64} x {f32,bf16}, streaming regime, at batch 4096 and 65536: GFLOP/s, frac of HBM roofline, % of MFMA peak.
  * reuse: the operand-reuse regimes -- B shared by the whole batch (stride_b = 0) and a blocked GEMM built from BRGEMM tiles
    (2-D batch, chain br = K/m, operands cache-resident): the only regime in which a percentage of MFMA peak is the binding roofline.
  * mfma_busy: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs) per workload from the committed PMC pass (profiles/), null if none.
  * cpu_baseline: the reference's own JIT kernel (oracle/_ref, kind "reference") or the C restatement (kind "port") on this box's host cores.
  * tpp: SURVEY 8 rows a11 (TPPs: copy, transpose, NORM -> VNNI2, config #5's un-fused bias-add + ReLU chain, row / column reductions, column gather), f1 (a five-node
    matrix equation) and f2 (a dense packed GEMM), each as [fraction of the HBM roofline, GB/s of the reference's CPU kernel on one host core], every GPU result
    checked against the oracle (tools/tpp_group.py).
  * configs: BASELINE configs #3 (packed CSR A-sparse 35x35 @15 % and @10 %, FsSpMDM), #4 (bf16 BCSC 2:8) and #5 (bf16 64^3 + bias + ReLU, one
    GPU's shard of 2^17 problems), each as [fraction of the HBM roofline, GFLOP/s of the reference's CPU kernel on one host core], every GPU
    result checked against the oracle; variantB: config #2 as ONE BRGEMM with br = 4096.

N > 1: one process per GPU, every rank owns its own batch (weak scaling, no data-path collective); time = max over ranks between
two barriers; value = total flops / time.  The ranks come from the launcher's environment (torch.distributed.run sets WORLD_SIZE / RANK /
LOCAL_RANK) -- or, when the plain command `python bench.py --gpus N ...` is run WITHOUT a launcher, bench.py starts N ranks itself
(spawn_ranks: the same torch.distributed.run command line the driver uses, rendezvous on 127.0.0.1) and relays their output, so
`--gpus N` means N ranks either way.  BENCH_DRY=1 is the plumbing check of exactly that on a box without GPUs (gloo, no kernel is
launched, `value` 0, "dry": true): tests/test_bench_contract_cpu.py.
`--config 5`: BASELINE configs[4] instead -- 2^20 bf16 64^3 problems with fused column-bias + ReLU, split over the ranks with
libxsmm_hip_shard_range (strong scaling), compute leg and (with --gather) the result gather to rank 0 timed separately.
"""
import argparse
import ctypes as C
import glob
import json
import math
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
MFMA_PEAK_TF = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0, "f64": 78.6}      # f64: v_mfma_f64_16x16x4_f64, 32 flop / clk / SIMD
L3_BYTES = 256 * 2 ** 20


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=[2, 5], help="BASELINE config: 2 = f32 32^3 batch 4096 (headline), 5 = bf16 64^3 fused, 2^20 problems sharded")
    ap.add_argument("--m", type=int, default=32, help="m = n = k")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--br", type=int, default=1, help="batch-reduce count of every problem")
    ap.add_argument("--beta", type=int, default=0, choices=[0, 1])
    ap.add_argument("--fused", type=int, default=0, help="1: bf16 column-bias + ReLU epilogue")
    ap.add_argument("--sets", type=int, default=0, help="distinct input sets to rotate over (0: auto, > 2x L3)")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="minimum length of the timed region")
    ap.add_argument("--total", type=int, default=1 << 20, help="--config 5: problems over all ranks")
    ap.add_argument("--gather", action="store_true", help="--config 5: also time the gather of C to rank 0 (RCCL)")
    ap.add_argument("--no-sweep", action="store_true", help="headline only (profiling passes)")
    ap.add_argument("--no-l3", action="store_true", help="skip the Infinity-Cache-resident secondary leg")
    ap.add_argument("--eager", action="store_true", help="plain launches instead of hipGraph replays in the timed regions (counter-collection passes)")
    ap.add_argument("--manifest", default="", help="write the execution order of the launches (label, kernel, counts) to this JSON file")
    ap.add_argument("--only", default="", help="measure only this sweep/reuse entry (profiling passes), e.g. reuse:f32_m32_blocked")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs #3 / #4 / #5 legs")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where the full record goes (the stdout line is the compact one)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--cpu-threads", type=int, default=-1, help="threads of the all-core CPU leg (-1: one per usable core, 0/1: skip)")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) under torch.distributed.run -- the command line the driver itself uses for
    N > 1 -- relay their output and make sure rank 0's JSON line is the LAST line on stdout.  Returns the launcher's exit code."""
    import socket
    import subprocess
    if os.environ.get("BENCH_DRY") != "1" and "BENCH_DEVICE" not in os.environ:
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} visible GPUs, this box has {have} (no rank is started: a rank without a device of its own would "
                  f"measure nothing; BENCH_DEVICE=0 BENCH_BACKEND=gloo shares one device between the ranks for plumbing checks)", file=sys.stderr)
            return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: starting " + " ".join(cmd), file=sys.stderr)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for text in proc.stdout:
        if line is not None:                # something followed the last candidate: it was not the last line, let it through
            sys.stdout.write(line)
            line = None
        if text.startswith("{") and '"metric"' in text:
            line = text
        else:
            sys.stdout.write(text)
    rc = proc.wait()
    sys.stdout.flush()
    if line is not None:
        sys.stdout.write(line if line.endswith("\n") else line + "\n")
        sys.stdout.flush()
    return rc


def dry_main(args):
    """BENCH_DRY=1: the rank / barrier / max-over-ranks / one-line plumbing of the N > 1 path with NO device and NO kernel (CPU-only boxes: the contract test).
    Nothing is computed and nothing is measured -- `value` is 0 and the line says "dry": true."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(os.environ.get("BENCH_BACKEND", "gloo"))
    from libxsmm_amd import parallel
    total = args.total if args.config == 5 else args.batch * world
    b, e = parallel.shard_range(total, world, rank) if args.config == 5 else (rank * args.batch, (rank + 1) * args.batch)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass                                 # a step of the dry run launches nothing
    elapsed = time.perf_counter() - t0
    owned = float(e - b)
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed, owned], dtype=torch.float64)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        elapsed, owned = float(t[0]), float(t[1])
    out = None
    if rank == 0:
        out = {"metric": f"GFLOP/s, batched stride-BRGEMM m=n=k={args.m} {args.dtype}", "value": 0.0, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 6), "higher_is_better": True, "scaling": "strong" if args.config == 5 else "weak", "vs_baseline": None,
               "dtype": args.dtype, "data": "none", "dry": True, "problems_owned_by_all_ranks": int(owned), "rccl_ranks": 0,
               "config": {"workload": "DRY RUN: ranks, barriers and the one-line output only; no device, no kernel, nothing measured", "per_gpu_batch": args.batch}}
    finish(dist, out, None)


def gen_values(n, bf16, dev, gen):
    """Reference-style data [samples/xgemm/gemm_kernel.c:837-865]: multiples of 0.1 in [-0.4, 0.5]; bf16 by truncation (bf16 = "f16": IEEE halves, RNE)."""
    out = torch.empty(n, dtype=torch.float64 if bf16 == "f64" else (torch.int16 if bf16 else torch.float32), device=dev)
    if os.environ.get("BENCH_ZERO_DATA") == "1":      # counter passes only (tools/macro_counters.sh): the same launches on all-zero operands -- what the matrix pipe clocks to without data toggling
        return out.zero_()
    step = 1 << 26
    for o in range(0, n, step):
        c = min(step, n - o)
        if bf16 == "f64":
            out[o:o + c] = torch.randint(-4, 6, (c,), generator=gen, device=dev, dtype=torch.int32).double() / 10
            continue
        v = torch.randint(-4, 6, (c,), generator=gen, device=dev, dtype=torch.int32).float() / 10
        if bf16 == "f16":
            out[o:o + c] = v.to(torch.float16).view(torch.int16)
        else:
            out[o:o + c] = (v.view(torch.int32) >> 16).to(torch.int16) if bf16 else v
    return out


class Workload:
    """mode 'stream': `batch` independent problems, each with its own A/B chain of `br` blocks and its own C.
    mode 'shared_b': the same, but one B chain shared by the whole batch (stride_b = 0).
    mode 'blocked': a blocked GEMM -- grid (ni, nj) of C tiles, C(i,j) = sum_r A(i,r) B(r,j), one 2-D batched launch."""

    def __init__(self, api, dev, dtype="f32", m=32, batch=4096, br=1, beta=0, fused=0, mode="stream", grid=None, nsets=0, seed=555, hint=None, tag=""):
        self.api, self.dev, self.dtype, self.m, self.br, self.beta, self.fused, self.mode, self.tag = api, dev, dtype, m, br, beta, fused, mode, tag
        self.bf16 = "f16" if dtype == "f16" else dtype == "bf16"       # truthy: 16-bit operands in VNNI-2 layout
        self.f64 = dtype == "f64"
        es = 2 if self.bf16 else (8 if self.f64 else 4)
        self.es = es
        kind = "f64" if self.f64 else self.bf16                        # what gen_values makes
        blk = m * m * es
        self.blk = blk
        if mode == "blocked":
            self.ni, self.nj = grid
            batch = self.ni * self.nj
            na, nb = self.ni * br, self.nj * br
        else:
            na, nb = batch * br, (br if mode == "shared_b" else batch * br)
        self.batch = batch
        self.flops_per_step = 2.0 * m * m * m * br * batch
        self.alg_bytes_per_step = float(na + nb) * blk + float(batch) * blk * (1 + beta) + (m * es if fused else 0)
        set_bytes = (na + nb + batch) * blk
        if nsets <= 0:
            nsets = 1 if mode == "blocked" else max(2, int(math.ceil(2.2 * L3_BYTES / set_bytes)))
        self.nsets = nsets
        # libxsmm_hip_set_streaming_hint for this workload's launches: a rotation over more input sets than the Infinity Cache holds IS
        # "operands read once from HBM" (2) and says so; everything else leaves the decision to the library (0)
        self.hint = hint if hint is not None else (2 if (mode == "stream" and nsets > 1 and nsets * set_bytes > 2 * L3_BYTES) else 0)
        gen = torch.Generator(device=dev).manual_seed(seed)
        tdt = torch.int16 if self.bf16 else (torch.float64 if self.f64 else torch.float32)
        skew = int(os.environ.get("XAMD_BENCH_SKEW", "0")) // es          # experiment: B / C start `skew` / 2 `skew` bytes into their allocations (channel aliasing of same-sized arrays)
        self.A = [gen_values(na * m * m, kind, dev, gen) for _ in range(nsets)]
        self.B = [gen_values(nb * m * m + skew, kind, dev, gen)[skew:] for _ in range(nsets)]
        self.C = [torch.zeros(batch * m * m + 2 * skew, dtype=tdt, device=dev)[2 * skew:] for _ in range(nsets)]
        self.D = gen_values(m, self.bf16, dev, gen) if fused else None
        t = DT.F16 if dtype == "f16" else (DT.BF16 if self.bf16 else (DT.F64 if self.f64 else DT.F32))
        self.t = t
        self.comp = DT.F64 if self.f64 else DT.F32
        self.flags = (0 if beta else GEMM_FLAG.BETA_0) | (GEMM_FLAG.VNNI_A if self.bf16 else 0)
        self.shape = capi.gemm_shape(m, m, m, m, m, m, t, t, t, self.comp)
        self.cfg = capi.br_config(capi.BR_STRIDE, blk, blk, 0)
        if fused:
            self.handle = api.dispatch_brgemm_ext(self.shape, self.flags, 0, self.cfg, capi.argops_cp(m, capi.UNARY.RELU, 0), capi.postops_colbias(m, t))
        else:
            self.handle = api.dispatch_brgemm(self.shape, self.flags, 0, self.cfg)
        if not self.handle:
            raise RuntimeError("dispatch returned NULL")
        self.brc = C.c_ulonglong(br)
        self.params = []
        for s in range(nsets):
            p = capi.GemmExtParam() if fused else capi.GemmParam()
            p.a.primary, p.b.primary, p.c.primary = self.A[s].data_ptr(), self.B[s].data_ptr(), self.C[s].data_ptr()
            p.op.tertiary = C.addressof(self.brc)
            if fused:
                p.d.primary = self.D.data_ptr()
            self.params.append(p)
        self.sa = br * blk
        self.sb = 0 if mode == "shared_b" else br * blk
        self.sc = blk

    def step(self, s):
        p = self.params[s % self.nsets]
        if self.mode == "blocked":
            self.api.hip_gemm_batch_strided_2d(self.handle, C.byref(p), self.ni, self.nj, self.sa, self.sb, self.sc, self.ni * self.sc)
        elif self.fused:
            self.api.hip_gemm_ext_batch_strided(self.handle, C.byref(p), self.batch, self.sa, self.sb, self.sc, 0, 0)
        else:
            self.api.hip_gemm_batch_strided(self.handle, C.byref(p), self.batch, self.sa, self.sb, self.sc)

    def kernel(self):
        return self.api.hip_kernel_name(self.handle, 1).decode()

    def label(self):
        return f"{self.dtype}_m{self.m}_" + (f"b{self.batch}" if self.mode == "stream" else (f"sharedB_b{self.batch}" if self.mode == "shared_b" else "blocked" + self.tag))

    # ---- the oracle as the checker: a strided sample of the batch, same inputs -----------------------------------
    def verify(self, s=0, samples=32):
        from oracle import pyoracle
        orc = pyoracle.oracle()
        m, br, mm = self.m, self.br, self.m * self.m
        npdt = np.uint16 if self.bf16 else (np.float64 if self.f64 else np.float32)
        idx = sorted(set(int(x) for x in np.linspace(0, self.batch - 1, samples)))
        A, B, Cg = self.A[s], self.B[s], self.C[s]
        flags = self.flags | GEMM_FLAG.BATCH_REDUCE_STRIDE | (GEMM_FLAG.USE_XGEMM_EXT_ABI if self.fused else GEMM_FLAG.USE_XGEMM_ABI)
        desc = pyoracle.GemmDesc(m, m, m, m, m, m, self.t, self.t, self.t, self.comp, flags, self.blk, self.blk, 1 if self.fused else 0, 1 if self.fused else 0)
        d_host = self.D.cpu().numpy().view(npdt) if self.fused else None
        worst = 0.0
        for e in idx:
            ia, ib = (e % self.ni, e // self.ni) if self.mode == "blocked" else (e, 0 if self.mode == "shared_b" else e)
            a = A[ia * br * mm:(ia + 1) * br * mm].cpu().numpy().view(npdt).copy()
            b = B[ib * br * mm:(ib + 1) * br * mm].cpu().numpy().view(npdt).copy()
            got = Cg[e * mm:(e + 1) * mm].cpu().numpy().view(npdt)
            ref = np.zeros(mm, dtype=npdt)          # beta = 1 accumulates over the steps: only beta = 0 workloads are verified
            p = capi.GemmExtParam() if self.fused else capi.GemmParam()
            brc = C.c_ulonglong(br)
            p.a.primary, p.b.primary, p.c.primary, p.op.tertiary = a.ctypes.data, b.ctypes.data, ref.ctypes.data, C.addressof(brc)
            if self.fused:
                p.d.primary = d_host.ctypes.data
            orc.gemm(p, desc)
            if self.dtype == "f16":
                r = ref.view(np.float16).astype(np.float64); g = got.view(np.float16).astype(np.float64)
            elif self.bf16:
                r = (ref.astype(np.uint32) << 16).view(np.float32).astype(np.float64); g = (got.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
            else:
                r = ref.astype(np.float64); g = got.astype(np.float64)
            den = float(np.sum(r * r)); num = float(np.sum((r - g) ** 2))
            worst = max(worst, math.sqrt(num / den) if den > 0 else math.sqrt(num))
        tol = 5e-3 if self.bf16 else (1e-12 if self.f64 else 1.2e-5)          # f64: this repo's own bound (tests/helpers.py); the rest: the reference driver's own bounds [samples/xgemm/gemm_kernel.c:5312-5414]
        return worst < tol, worst, len(idx)


def capture(work, launches, lanes=0):
    """`launches` back-to-back steps (rotating over the input sets) in one hipGraph, captured on a side stream.  lanes > 1: the steps are issued inside a
    pipeline section (libxsmm_hip_pipeline_begin / _end: the caller declares them independent), i.e. as `lanes` parallel branches of the graph."""
    api = work.api
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        api.hip_set_stream(side.cuda_stream)
        for i in range(2):
            work.step(i)
        side.synchronize()
        g.capture_begin()
        if lanes > 1:
            assert api.hip_pipeline_begin(lanes) == 0
        for i in range(launches):
            work.step(i)
        if lanes > 1:
            assert api.hip_pipeline_end() == 0
        g.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)
    return g


def estimate_step_seconds(work, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(3):
        work.step(i)
    e0.record()
    for i in range(n):
        work.step(i)
    e1.record(); torch.cuda.synchronize()
    return max(e0.elapsed_time(e1) * 1e-3 / n, 1e-6)


EAGER = False
REPLAY_SYNC = None  # N > 1: every rank replays its graph the SAME number of times (the largest any rank calibrated), so that "K steps" means one thing
MANIFEST = []      # execution order of the library launches, for tools/summarize_profiles.py (splits a kernel trace by workload)


def timed(work, steps, min_seconds, barrier=lambda: None, label=None, lanes=0):
    """A graph of G launches (G = a multiple of `steps` and of the rotation length) replayed R times so that the region lasts
    >= min_seconds, between two (barrier + synchronize) pairs.  Returns (wall seconds, launches timed, us per launch from HIP
    events recorded on the launch stream)."""
    work.api.hip_set_streaming_hint(work.hint)
    t_step = estimate_step_seconds(work)
    unit = steps * work.nsets // math.gcd(steps, work.nsets)
    per_graph = unit * max(1, min(int(0.02 / t_step) // unit, max(1, 2000 // unit)))
    while per_graph > 4000 and per_graph > steps:       # very long rotations: keep the graph bounded
        per_graph //= 2
    per_graph = max(per_graph, 1)
    replays = max(1, int(math.ceil(min_seconds / (per_graph * t_step))))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if EAGER:
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for i in range(per_graph * replays):
            work.step(i)
        e1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        n = per_graph * replays
        executed = int(work.api.hip_launch_count(1))
    else:
        g = capture(work, per_graph, lanes)
        g.replay(); torch.cuda.synchronize()              # untimed: the first replay uploads the graph
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(); g.replay(); c1.record(); torch.cuda.synchronize()        # untimed calibration: how long one replay really takes
        replays = max(1, int(math.ceil(min_seconds / max(c0.elapsed_time(c1) * 1e-3, 1e-6) * 1.05)))
        if REPLAY_SYNC is not None:
            replays = REPLAY_SYNC(replays)
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(replays):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        n = per_graph * replays
        # launches the GPU executed for this workload since the last manifest entry: the eager ones + the two untimed replays (upload,
        # calibration) + the timed replays; the capture itself went through the launch counter once without executing
        executed = int(work.api.hip_launch_count(1)) + per_graph + n
    work.api.hip_set_streaming_hint(0)
    MANIFEST.append({"label": label or work.label(), "kernel": work.kernel(), "launches_executed": executed, "launches_timed": n, "streaming_hint": work.hint,
                     "kernels_per_launch": getattr(work, "kernels_per_launch", 1), "lanes": lanes,
                     "algorithmic_bytes_per_launch": int(work.alg_bytes_per_step), "flops_per_launch": work.flops_per_step, "dtype": work.dtype,
                     "us_per_launch_events": e0.elapsed_time(e1) * 1e3 / n})
    return t1 - t0, n, e0.elapsed_time(e1) * 1e3 / n


def entry(work, steps, min_seconds, verify=True):
    """One sweep / reuse line."""
    work.api.hip_set_streaming_hint(work.hint)
    for i in range(3):
        work.step(i)
    torch.cuda.synchronize(); work.api.check()
    wall, n, us = timed(work, steps, min_seconds)
    work.api.check()
    tf = work.flops_per_step / (us * 1e-6) / 1e12
    gbs = work.alg_bytes_per_step / (us * 1e-6) / 1e9
    out = {"kernel": work.kernel(), "streaming_hint": work.hint, "us_per_launch": round(us, 3), "GFLOP/s": round(tf * 1e3, 1), "GB/s": round(gbs, 1),
           "frac_hbm": round(gbs / HBM_PEAK_GBS, 4), "pct_mfma_peak": round(100.0 * tf / MFMA_PEAK_TF[work.dtype], 2), "launches_timed": n}
    if verify:
        ok, err, cnt = work.verify(0)
        out["verified"] = bool(ok); out["normf_rel"] = float(f"{err:.3g}")
    return out


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


def cpu_baseline(m, dtype, br, beta, fused, seconds, nthreads):
    """Reference JIT (or C restatement) on this box's host cores: one thread, and every usable core at once; bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyoracle
    bf16, f64 = dtype == "bf16", dtype == "f64"
    es = 2 if bf16 else (8 if f64 else 4)
    t = DT.BF16 if bf16 else (DT.F64 if f64 else DT.F32)
    batch = 1024                                        # private operands, streamed like the GPU run
    rng = np.random.default_rng(555)
    raw = ((np.floor(rng.random(batch * br * m * m * 2) * 10) - 4) / 10).astype(np.float64 if f64 else np.float32)
    ab = (raw.view(np.uint32) >> 16).astype(np.uint16) if bf16 else raw
    A, B = ab[: batch * br * m * m].copy(), ab[batch * br * m * m:].copy()
    Cc = np.zeros(batch * m * m, dtype=np.uint16 if bf16 else (np.float64 if f64 else np.float32))
    Dd = ab[:m].copy()
    flags = (0 if beta else GEMM_FLAG.BETA_0) | (GEMM_FLAG.VNNI_A if bf16 else 0)
    shape = capi.gemm_shape(m, m, m, m, m, m, t, t, t, DT.F64 if f64 else DT.F32)
    cfg = capi.br_config(capi.BR_STRIDE, m * m * es, m * m * es, 0)
    brc = C.c_ulonglong(br)
    ptype = capi.GemmExtParam if fused else capi.GemmParam

    def make_param(a_, b_, c_):
        q = ptype()
        q.a.primary, q.b.primary, q.c.primary, q.op.tertiary = a_.ctypes.data, b_.ctypes.data, c_.ctypes.data, C.addressof(brc)
        if fused:
            q.d.primary = Dd.ctypes.data
        return q
    p = make_param(A, B, Cc)
    flops = 2.0 * m * m * m * br * batch
    if pyoracle.have_reference():
        ref = pyoracle.reference()
        if fused:
            h = ref.dispatch_brgemm_ext(shape, flags, 0, cfg, capi.argops_cp(m, capi.UNARY.RELU, 0), capi.postops_colbias(m, t))
            tfn = ref.lib.xref_time_gemm_ext_batch
        else:
            h = ref.dispatch_brgemm(shape, flags, 0, cfg)
            tfn = ref.lib.xref_time_gemm_batch
        if h:
            sa, sc = br * m * m * es, m * m * es
            t1 = tfn(h, C.byref(p), batch, sa, sa, sc, 5)
            reps = max(5, int(seconds / max(t1 / 5, 1e-9)))
            dt = tfn(h, C.byref(p), batch, sa, sa, sc, reps)
            out = {"value": round(flops * reps / dt / 1e9, 2), "unit": "GFLOP/s", "cores": 1, "kind": "reference",
                   "sample": f"reference JIT ({ref.lib.xref_get_target_arch().decode()}) kernel, {batch} problems x {reps} reps, private operands, 1 thread, {dt:.1f} s"}
            # the same kernel on every usable core at once (one thread per core, private operands): the caller's OpenMP loop of the
            # reference [samples/xgemm/gemm_kernel.c:4063-4066]; ctypes releases the GIL for the duration of each timing call
            if nthreads > 1:
                import threading
                sets = []
                for _ in range(nthreads):
                    a_, b_, c_ = A.copy(), B.copy(), Cc.copy()
                    sets.append((a_, b_, c_, make_param(a_, b_, c_)))

                def run_all(r):
                    times = [0.0] * nthreads

                    def work(i):
                        times[i] = tfn(h, C.byref(sets[i][3]), batch, sa, sa, sc, r)
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
                _, wall = run_all(reps_mt)
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


def single_call_latency():
    """The reference's calling pattern -- ONE small GEMM per call, the loop in the caller -- timed by a plain C program (examples/loop_driver.c, built by
    __graft_entry__.build()): microseconds per f32 32^3 call when every call blocks (the default, the reference's semantics), when calls are stream-ordered,
    and when consecutive calls are coalesced into one batched launch (libxsmm_hip_set_async(2)); wall clock around a loop of 4096 calls plus the final sync."""
    exe = os.path.join(ROOT, "libxsmm_amd", "lib", "loop_driver")
    if not os.path.exists(exe):
        return None
    import subprocess
    out = {}
    for mode in ("sync", "async", "coalesce"):
        try:
            r = subprocess.run([exe, "32", "4096", mode, "3", "f32"], capture_output=True, text=True, timeout=120)
            rec = json.loads(r.stdout.strip().splitlines()[-1])
            out[mode] = {"us_per_call": rec["us_per_call"], "GFLOP/s": rec["GFLOPs"], "launches_per_4096_calls": rec["launches_per_rep"], "bit_identical_to_batched_launch": rec["bit_identical"]}
        except Exception as e:                       # a side figure: never fails the bench
            out[mode] = {"error": str(e)[:100]}
    out["note"] = "examples/loop_driver.c: for (i < 4096) kernel(&param_i); libxsmm_hip_sync(); f32 32^3, device operands; the GPU is shared with this process while it runs"
    return out


def sharded_c_abi():
    """SURVEY 8(e) for C hosts: examples/sharded_driver.c (built by __graft_entry__.build()) -- ONE process, ONE thread, the batch cut by libxsmm_hip_shard_range into one
    block per shard, every shard's operands on the shard's device, one libxsmm_hip_gemm[_ext]_batch_strided_sharded call launches them all and gathers C on device 0.
    Shard s runs on device s % device_count: on a one-GPU box the four shards are virtual (a stream and scratch each), on an 8-GPU node the same command spreads.
    Reported: milliseconds per sharded launch INCLUDING the gather, and that the gathered C equals the unsharded launch bit for bit."""
    exe = os.path.join(ROOT, "libxsmm_amd", "lib", "sharded_driver")
    if not os.path.exists(exe):
        return None
    import subprocess
    out = {}
    for key, argv in (("f32_m32_b4096_x4", ["32", "4096", "4", "f32", "20"]), ("c5_bf16fused_m64_b32768_x4", ["64", "32768", "4", "bf16fused", "5"])):
        try:
            r = subprocess.run([exe] + argv, capture_output=True, text=True, timeout=300)
            rec = json.loads(r.stdout.strip().splitlines()[-1])
            out[key] = {"ms_per_sharded_launch_with_gather": rec["ms_per_sharded_launch_with_gather"], "GFLOP/s": rec["GFLOPs"], "shards": rec["shards"], "devices": rec["devices"],
                        "bit_identical_to_unsharded_launch": rec["bit_identical"], "rc": rec["rc"]}
        except Exception as e:                       # a side figure: never fails the bench
            out[key] = {"error": str(e)[:100]}
    return out


def stale_profile_rows(measured):
    """Every workload of this run against the committed rocprofv3 kernel-duration table (profiles/rNN_bench_kernel_stats.csv, the latest round): a row that is
    missing, names another kernel, or whose average duration is more than 15 % away from what this run measured is reported -- the driver line's numbers must be
    reproducible from profiles/ (round 3 shipped a variant-B row that predated the kernel's last change).  `measured`: {label: (kernel, us_per_launch)}."""
    import csv
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")))
    if not files:
        return None
    rows = {}
    try:
        for r in csv.DictReader(open(files[-1])):
            rows[r["workload"]] = (r["kernel"], float(r["avg_us"]), float(r.get("events_us_in_process") or 0.0))
    except Exception:
        return None
    stale = []
    # names the library reports for a template instance of another kernel / for a launch of several kernels (the table holds the first symbol of the launch)
    alias = {"bcsc_mfma_f32_stream_kernel": "bcsc_mfma_bf16_stream_kernel", "bcsc_mfma_f32_stream_full_kernel": "bcsc_mfma_bf16_stream_full_kernel", "gemm_fp8c8_stream_kernel": "gemm_fp8_stream_kernel", "gemm_bf32_stream_kernel": "gemm_f32_stream_kernel",
             "gemm_bitmask_reg_kernel": "bitmask_prepass_kernel", "gemm_i4_stream_kernel": "gemm_i8_stream_kernel", "gemm_i2_stream_kernel": "gemm_i8_stream_kernel",
             "gemm_i1_stream_kernel": "gemm_i8_stream_kernel", "gemm_bf16_wgp_kernel": "gemm_wgp16_kernel", "gemm_f16_wgp_kernel": "gemm_wgp16_kernel", "gemm_bf16_w64_kernel": "gemm_16bit_w64_kernel", "gemm_f16_w64_kernel": "gemm_16bit_w64_kernel",
             "gemm_8bit_wgp_kernel": "gemm_wgp8_kernel", "gemm_w8_wgp_kernel": "gemm_wgp16_kernel", "reduce_vec_kernel": "reduce_combine_kernel"}        # (the big column reduction is two kernels per call: partial sums, then their combination)
    for label, (kernel, us) in measured.items():
        if label not in rows:
            stale.append(f"{label}: no row"); continue
        k, avg, ev = rows[label]
        base = lambda n: n.replace("xamd::", "").split("<")[0].split("(")[0].split("+")[0].strip()      # noqa: E731
        if base(kernel) and base(k) != base(kernel) and base(k) != alias.get(base(kernel), ""):
            stale.append(f"{label}: row is {base(k)}, ran {base(kernel)}"); continue
        if us < 20.0:
            continue                                   # a launch of a few microseconds runs 1.5 - 2 x slower under the profiler: the row only has to name the kernel
        ref = ev if ev > 0 else avg                    # event time of the profiled run where the table has it (several kernels per launch: their sum)
        if us > 0 and abs(ref - us) / us > 0.15:
            stale.append(f"{label}: row {ref:.2f} us, measured {us:.2f} us")
    return {"table": os.path.relpath(files[-1], ROOT), "stale": stale}


def committed_counters(kernel, alg_bytes, label):
    """HBM traffic and MFMA-busy for this workload from the committed PMC passes (rocprofv3 --pmc cannot run inside this process:
    tools/profile_paths.sh runs THIS command under it in separate passes, tools/summarize_profiles.py distils profiles/)."""
    traffic = src = busy = busy_src = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            for w in json.load(open(f))["workloads"]:
                if w["algorithmic_bytes_per_launch"] == int(alg_bytes) and w["kernel"].replace(" ", "") == kernel.replace(" ", ""):
                    traffic, src = w["traffic_bytes_per_launch"], os.path.relpath(f, ROOT)
                    break
        except Exception:
            continue
        if traffic is not None:
            break
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma_busy.json")), reverse=True):
        try:
            d = json.load(open(f))
            if label in d.get("workloads", {}):
                busy, busy_src = d["workloads"], os.path.relpath(f, ROOT)
                break
        except Exception:
            continue
    return traffic, src, busy, busy_src


def committed_copy_floor():
    """{name: best fraction of 8 TB/s} of the latest committed tools/copy_floor.hip run (profiles/r*_copy_floor.txt): what a kernel that ONLY moves a config's bytes --
    same read : write mix, same footprint -- reached on the GPU box of that round.  Not measured in this run (said so in the key that carries it)."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_copy_floor.txt")), reverse=True):
        best = {}
        try:
            for ln in open(f):
                if ln.startswith("{"):
                    r = json.loads(ln)
                    best[r["copy_floor"]] = max(best.get(r["copy_floor"], 0.0), float(r["frac_of_8TBs"]))
        except Exception:
            continue
        if best:
            short = {"c2_headline_f32_32x4096": "c2", "c5_bf16_64x131072": "c5", "bf16_32x65536": "bf16_32", "c3_csr_35x35_P65536": "c3_csr", "c3_fsspmdm_N2p20": "c3_fss", "c4_bcsc_8192": "c4"}
            return {"source": os.path.relpath(f, ROOT), "frac_of_8TBs": {short[k]: round(v, 3) for k, v in best.items() if k in short}}     # (the compact line is size-bound)
    return None


SWEEP = [(dt, m, b) for dt in ("f32", "bf16", "f64") for m in (16, 32, 64) for b in (4096, 65536)]      # f64 (round 4): v_mfma_f64_16x16x4_f64
# blocked GEMMs: (dtype, m, ni, nj, br, tag) -- f32: 2048^3 out of 16^3 tiles, 4096 x 4096 x 2048 out of 32^3 tiles, 4096^3 out of 64^3 tiles (macro tile 128 x 128:
# 256 .. 1024 workgroups); bf16 (macro tile 256 x 256): 4096^3 out of 16^3 / 32^3 / 64^3 tiles = ONE round of 256 workgroups, and 8192^3 out of 64^3 tiles = four rounds
BLOCKED = [("f32", 16, 128, 128, 128, ""), ("f32", 32, 128, 128, 64, ""), ("f32", 64, 64, 64, 64, ""),
           ("bf16", 16, 256, 256, 256, ""), ("bf16", 32, 128, 128, 128, ""), ("bf16", 64, 64, 64, 64, ""), ("bf16", 64, 128, 128, 128, "_8192"),
           ("f64", 32, 128, 128, 64, ""), ("f64", 64, 64, 64, 64, "")]      # f64: 4096 x 4096 x 2048 out of 32^3 tiles, 4096^3 out of 64^3 tiles (macro tile 128 x 128)
SHARED_B = [(dt, m, 65536) for dt in ("f32", "bf16") for m in (16, 32, 64)]
# the odd shapes LIBXSMM is known for (BASELINE configs[0] is one 23^3 f32 GEMM): same streaming regime, problems that are not whole tiles
RAGGED = [("f32", 23, 131072), ("f32", 23, 4096), ("f32", 13, 262144), ("f32", 40, 32768), ("f32", 72, 16384),
          ("bf16", 40, 65536), ("bf16", 72, 16384), ("bf16", 96, 8192)]      # round 5: ragged / several-tile 16-bit shapes (one problem per workgroup out of LDS, gemm_wgp16_kernels.hip)


def mfma_roof(api, dev):
    """What the matrix pipe of THIS chip sustains on the bench's operand values with nothing but MFMAs (libxsmm_hip_probe_mfma: register operands, one wave
    per SIMD, no LDS, no memory): the roof a blocked GEMM on the same data cannot exceed -- the chip clocks to its power budget, and the 2.5 PF bf16 figure
    assumes 2.4 GHz.  TFLOP/s per data type."""
    out = {}
    gen = torch.Generator(device=dev).manual_seed(555)
    for name, t, iters in (("bf16", DT.BF16, 6000), ("f32", DT.F32, 3000)):
        ops = gen_values(32768 if name == "bf16" else 16384, name == "bf16", dev, gen)       # 64 KiB of the bench's value distribution
        flop = C.c_double(0.0)
        for _ in range(2):
            api.hip_probe_mfma(t, ops.data_ptr(), iters, C.byref(flop))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            api.hip_probe_mfma(t, ops.data_ptr(), iters, C.byref(flop))
        e1.record(); torch.cuda.synchronize(); api.check()
        out[name] = round(5 * flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
    return out


PIPELINED = [("f32", 32, 4096), ("f32", 16, 4096), ("bf16", 16, 4096), ("bf16", 32, 4096), ("f32", 23, 4096), ("f32", 64, 4096), ("bf16", 64, 4096)]


def run_pipelined(api, dev, steps, min_seconds, lanes=4):
    """The small-launch regime with INDEPENDENT consecutive launches allowed to overlap (libxsmm_hip_pipeline_begin / _end, `lanes` internal streams,
    captured as parallel branches of the graph): a launch of 4096 small problems is one round of waves, a third of it fill and drain.  The strict
    one-launch-after-the-other figures stay the headline / sweep numbers; this is what a caller gets who declares the independence it has."""
    out = {"lanes": lanes}
    for dt, m, b in PIPELINED:
        try:
            set_bytes = 3 * b * m * m * (2 if dt == "bf16" else 4)
            nsets = max(2, int(math.ceil(2.2 * L3_BYTES / set_bytes)))
            nsets = (nsets + lanes - 1) // lanes * lanes          # launch k and launch k + nsets write the same C: they land on the same lane (ordered)
            w = Workload(api, dev, dt, m, b, nsets=nsets)
            api.hip_set_streaming_hint(w.hint)
            for i in range(3):
                w.step(i)
            torch.cuda.synchronize(); api.check()
            _, n, us = timed(w, steps, min_seconds, label=w.label() + f"_pipelined{lanes}", lanes=lanes)
            api.check()
            ok, err, _ = w.verify(0)
            gbs = w.alg_bytes_per_step / (us * 1e-6) / 1e9
            out[w.label()] = {"kernel": w.kernel(), "us_per_launch": round(us, 3), "GFLOP/s": round(w.flops_per_step / us / 1e3, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4),
                              "launches_timed": n, "input_sets_rotated": nsets, "verified": bool(ok)}
            del w
        except Exception as e:
            out[f"{dt}_m{m}_b{b}"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    return out


def variant_b(api, dev, br):
    """config #2 as ONE strided BRGEMM with a chain of `br` blocks: two kernels per call (slices of the chain reduced on chip, then across slices)"""
    w = Workload(api, dev, "f32", 32, 1, br=br)
    w.kernels_per_launch = 2
    w.kernel = lambda: api.hip_kernel_name(w.handle, 0).decode()           # a single call, not a batched launch
    if br > 8192:
        w.verify = lambda s=0, samples=32: None      # the oracle's own serial f32 chain of 2M terms is no sharper than the split sum: checked at br = 4096
    return w


def run_configs(api, dev, steps, min_seconds, cpu_seconds, with_cpu):
    """BASELINE configs #3, #4, #5 (one GPU) and config #2's variant B, measured like the headline (hipGraph replays, HIP events on the launch
    stream, inputs rotated past the Infinity Cache), every result checked against the oracle, the reference's CPU kernel timed beside each."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import workloads as wl
    wl.set_device(dev)
    cs = max(1.0, min(3.0, cpu_seconds))
    specs = [
        ("c3_csr15", lambda: wl.csr_asparse(api, 65536, 0.15), lambda: wl.cpu_csr(1024, 0.15, cs)),
        ("c3_csr10", lambda: wl.csr_asparse(api, 65536, 0.10), lambda: wl.cpu_csr(1024, 0.10, cs)),
        ("c3_fsspmdm", lambda: wl.fsspmdm(api, 2 ** 20, 0.15), lambda: wl.cpu_fsspmdm(49152, 0.15, cs)),
        # the same operator on N = 10^6 columns: with N = 2^20 the 35 rows of B and of C lie exactly 8 MiB apart and every wave's 70 streams fall on the same
        # HBM channels and banks (measured: f32 0.56 at 2^20 against 0.74 at 2^20 + 8704, f64 0.64 against 0.66-0.69) -- the caller's leading dimension, not the kernel
        ("c3_fsspmdm_n1e6", lambda: wl.fsspmdm(api, 1000000, 0.15), None),
        ("c4_bcsc_bf16", lambda: wl.bcsc(api, host_pattern=True), lambda: wl.cpu_bcsc(seconds=cs)),
        # config #4's f32 and 8-bit siblings: the same pattern and shape in the other two operand types of the reference's spmm_kernel driver
        ("c4_bcsc_f32", lambda: wl.bcsc(api, dtype="f32", host_pattern=True), None),
        ("c4_bcsc_u8i8", lambda: wl.bcsc(api, dtype="u8i8", host_pattern=True), None),
        ("c5_fused", lambda: Workload(api, dev, "bf16", 64, 2 ** 17, fused=1), lambda: wl.cpu_fused(seconds=cs)),
        ("variantB_f32_m32_br4096", lambda: variant_b(api, dev, 4096), None),
        ("variantB_f32_m32_br65536", lambda: variant_b(api, dev, 65536), None),
    ]
    out = {}
    for label, make, cpu in specs:
        try:
            w = make()
            for i in range(3):
                w.step(i)
            torch.cuda.synchronize(); api.check()
            _, n, us = timed(w, steps, min_seconds, label=label)
            api.check()
            gbs = w.alg_bytes_per_step / (us * 1e-6) / 1e9
            r = {"workload": getattr(w, "name", None) or w.label(), "kernel": w.kernel(), "us_per_launch": round(us, 3), "GFLOP/s": round(w.flops_per_step / us / 1e3, 1),
                 "GB/s": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(w.alg_bytes_per_step), "launches_timed": n,
                 "input_sets_rotated": w.nsets}
            if hasattr(w, "dense_equiv_flops"):
                r["dense_equiv_GFLOP/s"] = round(w.dense_equiv_flops / us / 1e3, 1)
            v = w.verify() if hasattr(w, "verify") else None
            if v is not None and len(v) == 3:
                v = v[:2]
            if v is not None:
                r["verified"] = bool(v[0]); r["normf_rel"] = float(f"{v[1]:.3g}")
            if cpu is not None and with_cpu:
                try:
                    from oracle import pyoracle
                    r["cpu_baseline"] = cpu() if pyoracle.have_reference() else None
                except Exception as e:          # the CPU leg must never take the GPU measurement down
                    r["cpu_baseline"] = {"error": repr(e)}
            out[label] = r
            del w
        except Exception as e:                  # a workload that cannot be built or run is reported, not hidden
            out[label] = {"error": repr(e)}
        torch.cuda.empty_cache()
    return out


def run_round4(api, dev, steps, min_seconds):
    """Round-4 workloads outside BASELINE's configs, measured like everything else here (hipGraph replays, HIP events, rotated inputs): 8-bit GEMMs of a shape that is
    not whole tiles on the masked matrix-core kernel, an 8-bit float GEMM with a result of its own type, 8-bit weights x bf16 activations, the bitmask-compressed-A GEMM and the NORM -> VNNI2
    transform with leading dimensions that are no multiples of eight.  Parity of each is the GPU test-suite's job (tests/test_gemm_gpu.py: SHAPES_I8 / SHAPES_FP8,
    test_more_gemm_types_bit_exact; tests/test_full_size_gpu.py: bitmask; tests/test_meltw_gpu.py: TRANSFORMS); here only the clock runs."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_paths as bp
    import workloads as wl
    from libxsmm_amd.capi import DT, GEMM_FLAG, UNARY
    wl.set_device(dev); bp.DEV = dev
    specs = [
        ("i8_m40", lambda: bp.brgemm_i8(api, 40, 2 ** 17, ua=False)),
        ("u8i8_m40", lambda: bp.brgemm_i8(api, 40, 2 ** 17, ua=True)),
        ("bf8_m40", lambda: bp.brgemm_form(api, 40, 2 ** 17, GEMM_FLAG.VNNI_A, a_dt=DT.BF8, c_dt=DT.F32, name="bf8 -> f32")),
        ("hf8c8_m64", lambda: bp.brgemm_form(api, 64, 2 ** 16, GEMM_FLAG.VNNI_A, a_dt=DT.HF8, c_dt=DT.HF8, name="hf8 -> hf8")),
        ("bf16_m40", lambda: bp.brgemm(api, 40, "bf16", 2 ** 16)),                                  # a ragged bf16 shape on the masked bf16 matrix-core kernel
        ("w8_bf8_m64", lambda: bp.brgemm_w8(api, 64, 2 ** 16, DT.BF8, True)),                     # 8-bit float weights (VNNI-2 pairs) x bf16 -> bf16
        ("w8_i8s_m64", lambda: bp.brgemm_w8(api, 64, 2 ** 16, DT.I8, False, DT.F32)),              # int8 weights with row scales x bf16 -> f32
        # round 5: 72^3 of the 8-bit families -- packed blocks of nine tiles, one problem per workgroup out of LDS (gemm_wgp8_kernel / gemm_wgp16_kernel<.., AK>)
        ("i8_m72", lambda: bp.brgemm_i8(api, 72, 2 ** 15, ua=False)),
        ("bf8_m72", lambda: bp.brgemm_form(api, 72, 2 ** 15, GEMM_FLAG.VNNI_A, a_dt=DT.BF8, c_dt=DT.F32, name="bf8 -> f32")),
        ("w8_bf8_m72", lambda: bp.brgemm_w8(api, 72, 2 ** 14, DT.BF8, True)),
        ("bitmaskA_8192x64", lambda: bp.bitmask_gemm(api, 8192, 64, 8192, 0.5)),
        ("vnni2_ld4090", lambda: bp.meltw_big(api, UNARY.TRANSFORM_NORM_TO_VNNI2, "NORM_TO_VNNI2 bf16", m=4090, in_dt=DT.BF16, out_dt=DT.BF16)),
    ]
    out = {}
    for label, make in specs:
        try:
            w = make()
            for i in range(3):
                w.step(i)
            torch.cuda.synchronize(); api.check()
            _, n, us = timed(w, steps, min_seconds, label=label)
            api.check()
            ab = getattr(w, "alg_bytes", None) or w.alg_bytes_per_step
            out[label] = {"workload": getattr(w, "name", None) or w.label(), "kernel": w.kernel(), "us_per_launch": round(us, 3), "frac_hbm": round(ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                          "algorithmic_bytes_per_launch": int(ab), "launches_timed": n}
            del w
        except Exception as e:               # a side group: reported, never fails the bench
            out[label] = {"error": repr(e)[:120]}
        torch.cuda.empty_cache()
    return out


def compact_line(full, detail_path):
    """The driver keeps an 8 KB tail of stdout: the line it parses carries the contract fields and ONE OR TWO numbers per secondary workload
    ([frac of HBM roofline, % of MFMA peak] for sweep / reuse / ragged, [frac, CPU GFLOP/s on one core] for the BASELINE configs)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "steps_timed", "warmup", "ms_per_step", "timed_region_s", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "verified", "pct_mfma_peak", "rccl_ranks", "gather_ms", "gather_GBs_into_root", "dry", "problems_owned_by_all_ranks")
    line = {k: full[k] for k in keep if k in full}
    cfg = full.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "kernel", "per_gpu_batch", "problems_per_gpu_rank0", "input_sets_rotated", "streaming_hint", "parallelism") if k in cfg}
    rf = full.get("roofline")
    if rf:
        line["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_us", "algorithmic_bytes_per_launch")}
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind") if k in cb}
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:64]
        if "all_cores" in cb:
            line["cpu_baseline"]["all_cores"] = {"value": cb["all_cores"]["value"], "cores": cb["all_cores"]["cores"]}
    sc = full.get("single_call_us")
    if sc:
        line["single_call_us"] = [sc.get(k, {}).get("us_per_call") for k in ("sync", "async", "coalesce")]       # [blocking, stream-ordered, coalesced] per f32 32^3 call
    sh = full.get("sharded_c_abi")
    if sh:     # [ms per sharded launch with gather, shards, devices, bit-identical to the unsharded launch] per workload of examples/sharded_driver.c (the C-ABI multi-device launcher)
        line["sharded_c_abi"] = {k: (None if "error" in v else [v["ms_per_sharded_launch_with_gather"], v["shards"], v["devices"], v["bit_identical_to_unsharded_launch"]]) for k, v in sh.items()}
    if full.get("profiles_check"):
        line["profiles_stale_rows"] = len(full["profiles_check"]["stale"])            # workloads whose committed rocprofv3 row does not reproduce this run (0 = all do)
    cb64 = full.get("cpu_baseline_f64")
    if cb64:
        line["cpu_baseline_f64"] = [cb64.get("value"), cb64.get("cores"), cb64.get("kind")]       # [GFLOP/s of the reference's f64 32^3 kernel, cores, kind]
    if full.get("configs"):
        line["configs"] = {}
        ok = True
        for k, r in full["configs"].items():
            if "error" in r:
                line["configs"][k] = None; ok = False
                continue
            cpu = r.get("cpu_baseline") or {}
            line["configs"][k] = [r["frac_hbm"], cpu.get("value")]
            ok = ok and r.get("verified", True)
        line["configs_verified"] = bool(ok)
        line["configs_fields"] = "[frac_hbm, cpu_GFLOPs_1core]"
    for grp in ("sweep", "reuse", "ragged"):
        if full.get(grp):
            line[grp] = {k: [round(r["frac_hbm"], 3), round(r["pct_mfma_peak"], 1)] for k, r in full[grp].items()}
    if any(full.get(g) for g in ("sweep", "reuse", "ragged")):
        line["sweep_fields"] = "[frac_hbm, pct_mfma_peak]"
        line["sweep_verified"] = all(r.get("verified", True) for g in ("sweep", "reuse", "ragged") for r in (full.get(g) or {}).values())
    if full.get("reuse"):
        ppr = {k: r["pct_of_power_roof"] for k, r in full["reuse"].items() if r.get("pct_of_power_roof") is not None}
        if ppr:
            line["pct_of_power_roof"] = ppr           # blocked entries: % of what back-to-back MFMAs on the same operand values sustain under the power budget (mfma_power_roof_TF)
    if full.get("round4"):
        line["round4"] = {k: r.get("frac_hbm") for k, r in full["round4"].items()}       # fractions of the HBM roofline; shapes / kernels in the detail record
    if full.get("tpp"):
        # [fraction of the HBM roofline, GB/s of the reference's CPU kernel on one host core (null: none)], every entry checked against the oracle
        line["tpp"] = {k: (None if "error" in r else [r.get("frac_hbm"), (r.get("cpu_baseline") or {}).get("GB/s")]) for k, r in full["tpp"].items()}
        line["tpp_verified"] = all(("error" not in r) and r.get("verified") is True for r in full["tpp"].values())
    if full.get("pipelined"):
        pl = full["pipelined"]
        line["pipelined"] = {k: (v if not isinstance(v, dict) else (round(v["frac_hbm"], 3) if "frac_hbm" in v else None)) for k, v in pl.items()}
        line["pipelined_verified"] = all(v.get("verified", False) for v in pl.values() if isinstance(v, dict))
    if full.get("effective_clock_GHz"):
        ec = full["effective_clock_GHz"]
        line["effective_clock_GHz_committed_profile"] = [ec.get("bf16_m64_blocked_8192"), ec.get("bf16_m64_blocked_8192_on_zeros"), ec.get("mfma_busy_frac")]   # [drivers' data, zeros, matrix pipe busy]: committed rocprofv3 pass
    for k in ("l3_resident_us", "without_streaming_hint_us", "mfma_power_roof_TF"):
        if full.get(k) is not None:
            line[k] = full[k]
    cf = committed_copy_floor()
    if cf:
        line["copy_floor_committed_profile"] = cf
    line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    return line


def run_config5(args, api, dev, rank, world, dist, barrier):
    """BASELINE configs[4]: 2^20 bf16 64^3 BRGEMMs with fused column-bias + ReLU, problems split by batch index."""
    b, e = C.c_size_t(0), C.c_size_t(0)
    api.hip_shard_range(args.total, 1, world, rank, C.byref(b), C.byref(e))
    mine = e.value - b.value
    work = Workload(api, dev, "bf16", 64, mine, br=1, beta=0, fused=1, mode="stream", nsets=1 if mine * 24576 > 2.2 * L3_BYTES else 0, seed=555 + rank)
    for i in range(max(1, min(args.warmup, 5))):
        work.step(i)
    torch.cuda.synchronize(); api.check()
    elapsed, n, us = timed(work, args.steps, args.min_seconds, barrier)
    api.check()
    ok, err, cnt = work.verify(0)
    gather_ms = None
    if args.gather and dist is not None:
        # the only collective of the path: the final result gather (RCCL over xGMI), timed on its own
        from libxsmm_amd import parallel
        shard = work.C[0].view(torch.uint8).view(mine, -1)              # bytes: RCCL has no 16-bit integer type; one row per problem
        # RCCL: point-to-point, every source straight to rank 0 over its own link; any other backend (the one-GPU plumbing test under gloo): the C ABI's
        # IPC gather (libxsmm_hip_ipc_export / libxsmm_hip_gather_shards) -- the root pulls every shard with one device copy per source
        gather = parallel.gather_to_root if dist.get_backend() == "nccl" else parallel.gather_shards_ipc
        for _ in range(2):
            gather(shard, args.total, root=0)
        torch.cuda.synchronize(); barrier()
        t0 = time.perf_counter()
        full_c = gather(shard, args.total, root=0)
        torch.cuda.synchronize(); barrier()
        gather_ms = (time.perf_counter() - t0) * 1e3
        if rank == 0:                                                   # the assembled result carries this rank's shard where shard_range says it lies
            ok = ok and full_c is not None and full_c.shape[0] == args.total and bool(torch.equal(full_c[b.value:e.value], shard))
    if dist is not None:
        t = torch.tensor([elapsed, float(n), 0.0 if ok else 1.0], dtype=torch.float64, device=dev)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, bad = float(tmax[0].item()), float(tmax[2].item())
        tmin = t.clone(); dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        n = int(tmin[1].item())            # every rank times the same number of launches unless its estimate differed: count the fewest
        ok = bad == 0.0
    if rank != 0:
        return None
    flops = 2.0 * 64 ** 3 * args.total * n
    value = flops / elapsed / 1e9
    per_gpu_bytes = work.alg_bytes_per_step
    gbs = per_gpu_bytes / (us * 1e-6) / 1e9
    out = {"metric": "GFLOP/s, batched stride-BRGEMM m=n=k=64 bf16 + fused colbias+ReLU (BASELINE config 5)", "value": round(value, 1), "unit": "GFLOP/s",
           "n_gpus": world, "steps": args.steps, "steps_timed": n, "warmup": args.warmup, "ms_per_step": round(elapsed / n * 1e3, 5), "timed_region_s": round(elapsed, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"stride-BRGEMM bf16 m=n=k=64 br=1 beta=0 + column bias + ReLU, {args.total} problems split over {world} rank(s) by libxsmm_hip_shard_range",
                      "kernel": work.kernel(), "problems_per_gpu_rank0": mine, "parallelism": f"batch-split x{world}, no data-path collective"},
           "rccl_ranks": world if dist is not None else 0, "verified": bool(ok), "verify_normf_rel": float(f"{err:.3g}"),
           "pct_mfma_peak": round(100.0 * value / world / 1e3 / MFMA_PEAK_TF["bf16"], 2),
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                        "kernel_us": round(us, 3), "algorithmic_bytes_per_launch": int(per_gpu_bytes), "note": "rank 0's launch; every rank runs the same shard size +-1"}}
    if gather_ms is not None:
        out["gather_ms"] = round(gather_ms, 3)
        out["gather_GBs_into_root"] = round((args.total - mine) * 8192 / (gather_ms * 1e-3) / 1e9, 1)
    return out


def main():
    global EAGER
    args = parse()
    EAGER = args.eager
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:            # the plain command: no launcher set the ranks up, so bench.py does
        sys.exit(spawn_ranks(args.gpus))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: WARNING: --gpus {args.gpus} but the launcher started {os.environ['WORLD_SIZE']} rank(s): the launcher decides", file=sys.stderr)
    if os.environ.get("BENCH_DRY") == "1":
        dry_main(args)
        return
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # One process per GPU: LOCAL_RANK is the device.  BENCH_DEVICE / BENCH_BACKEND exist for the plumbing test on a ONE-GPU box (tests/test_parallel_gloo.py:
    # two ranks share device 0 behind a gloo group -- RCCL refuses two ranks on one device -- and the result gather goes through the C ABI's IPC gather).
    devidx = int(os.environ.get("BENCH_DEVICE", local))
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(devidx)
    dev = torch.device("cuda", devidx)
    local = devidx
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":      # BENCH_FORCE_DIST: exercise the RCCL code path with one rank (1-GPU boxes)
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    if dist is not None:
        def _same_replays(r):
            t = torch.tensor([float(r)], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return int(t.item())
        global REPLAY_SYNC
        REPLAY_SYNC = _same_replays

    api = capi.load()
    api.hip_set_device(local)
    api.hip_set_stream(torch.cuda.current_stream().cuda_stream)      # stream-ordered launches on torch's stream
    nthreads = usable_cpus() if args.cpu_threads < 0 else args.cpu_threads

    if args.config == 5:
        out = run_config5(args, api, dev, rank, world, dist, barrier)
        if rank == 0 and not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(64, "bf16", 1, 0, 1, args.cpu_seconds, nthreads)
        finish(dist, out if rank == 0 else None, args.detail)
        return

    only = args.only
    if only:                 # selected sweep / reuse entries under a profiler: no headline, no CPU leg
        api.hip_launch_count(1)
        results = {}
        for item in only.split(","):
            kind, label = item.split(":")
            found = None
            if kind == "sweep":
                for dt, m, b in SWEEP:
                    if f"{dt}_m{m}_b{b}" == label:
                        found = Workload(api, dev, dt, m, b)
            else:
                for dt, m, b in SHARED_B:
                    if f"{dt}_m{m}_sharedB_b{b}" == label:
                        found = Workload(api, dev, dt, m, b, mode="shared_b")
                for dt, m, ni, nj, br, tag in BLOCKED:
                    if f"{dt}_m{m}_blocked{tag}" == label:
                        found = Workload(api, dev, dt, m, 0, br=br, mode="blocked", grid=(ni, nj), tag=tag)
            if found is None:
                raise SystemExit(f"unknown entry {item}")
            results[label] = entry(found, args.steps, args.min_seconds)
            del found; torch.cuda.empty_cache()
        print(json.dumps({"only": only, "results": results}))
        if args.manifest:
            json.dump({"command": " ".join(sys.argv), "entries": MANIFEST}, open(args.manifest, "w"), indent=1)
        return

    work = Workload(api, dev, args.dtype, args.m, args.batch, br=args.br, beta=args.beta, fused=args.fused, nsets=args.sets)
    api.hip_launch_count(1)
    api.hip_set_streaming_hint(work.hint)
    for i in range(args.warmup):
        work.step(i)
    torch.cuda.synchronize()
    api.check()
    elapsed, n_timed, kernel_us = timed(work, args.steps, args.min_seconds, barrier)
    api.check()
    headline_kernel, headline_hint = work.kernel(), work.hint
    verified, verr, vcnt = (work.verify(0) if args.beta == 0 else (None, 0.0, 0))
    if dist is not None:
        t = torch.tensor([elapsed, -float(n_timed), 0.0 if verified in (True, None) else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, n_timed = float(t[0].item()), int(-t[1].item())
        if verified is not None:
            verified = t[2].item() == 0.0
    # secondary measurement: the same set every step (Infinity-Cache resident), not the headline
    l3_us = auto_us = None
    if not args.no_l3:
        # the same rotation without the declaration (hint 0: the library decides by launch size and, since round 6, by the thread's recent operand sets together)
        work.hint = 0
        _, _, auto_us = timed(work, args.steps, min(args.min_seconds, 0.2), label=work.label() + "_hint0")
        l3 = Workload(api, dev, args.dtype, args.m, args.batch, br=args.br, beta=args.beta, fused=args.fused, nsets=1)
        l3_elapsed, l3_n, l3_us = timed(l3, args.steps, min(args.min_seconds, 0.2), label=work.label() + "_l3resident")
        del l3

    sweep, reuse, ragged = {}, {}, {}
    if rank == 0 and world == 1 and not args.no_sweep:
        quick = min(args.min_seconds, 0.15)
        for dt, m, b in RAGGED:
            w = Workload(api, dev, dt, m, b)
            ragged[w.label()] = entry(w, args.steps, quick)
            del w; torch.cuda.empty_cache()
        for dt, m, b in SWEEP:
            w = Workload(api, dev, dt, m, b)
            sweep[w.label()] = entry(w, args.steps, quick)
            del w; torch.cuda.empty_cache()
        for dt, m, b in SHARED_B:
            w = Workload(api, dev, dt, m, b, mode="shared_b")
            reuse[w.label()] = entry(w, args.steps, quick)
            del w; torch.cuda.empty_cache()
        for dt, m, ni, nj, br, tag in BLOCKED:
            w = Workload(api, dev, dt, m, 0, br=br, mode="blocked", grid=(ni, nj), tag=tag)
            r = entry(w, args.steps, quick)
            r["gemm"] = f"{ni * m}x{nj * m}x{br * m} as {ni}x{nj} tiles of {m}^3, br={br}"
            reuse[w.label()] = r
            del w; torch.cuda.empty_cache()
    configs, roof = {}, None
    if rank == 0 and world == 1 and not args.no_sweep and not args.no_configs:
        configs = run_configs(api, dev, args.steps, min(args.min_seconds, 0.15), args.cpu_seconds, not args.no_cpu_baseline)
    round4 = {}
    if rank == 0 and world == 1 and not args.no_sweep and not args.no_configs:
        round4 = run_round4(api, dev, args.steps, min(args.min_seconds, 0.15))
    tpp = {}
    if rank == 0 and world == 1 and not args.no_sweep and not args.no_configs:
        # SURVEY 8 rows a11 / f1 / f2 in the driver-visible line (round-4 review item 4): tools/tpp_group.py -- measured like the headline, verified against the oracle
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import tpp_group
        tpp = tpp_group.run(api, dev, args.steps, min(args.min_seconds, 0.15), args.cpu_seconds, not args.no_cpu_baseline, timed)
    pipelined = None
    if rank == 0 and world == 1 and not args.no_sweep:
        pipelined = run_pipelined(api, dev, args.steps, min(args.min_seconds, 0.15), lanes=int(os.environ.get("BENCH_LANES", "8")))
        roof = mfma_roof(api, dev)
        for r in reuse.values():           # blocked entries next to what the pipe sustains on this data, on this chip, today
            if "gemm" in r:
                dt = "bf16" if r["kernel"].startswith("gemm_bf16") else ("f64" if r["kernel"].startswith("gemm_f64") else "f32")
                if dt in roof:
                    r["pct_of_power_roof"] = round(100.0 * r["GFLOP/s"] / 1e3 / roof[dt], 1)
    if dist is not None:
        dist.barrier()

    if rank == 0:
        traffic, traffic_src, busy, busy_src = committed_counters(headline_kernel, work.alg_bytes_per_step, work.label())
        if traffic is None:
            print(f"bench.py: WARNING: no committed PMC pass (profiles/r*_pmc_traffic.json) matches kernel {headline_kernel!r} with "
                  f"{int(work.alg_bytes_per_step)} algorithmic bytes per launch: roofline.traffic is null (re-run tools/profile_paths.sh)", file=sys.stderr)
        value = work.flops_per_step * n_timed * world / elapsed / 1e9
        gbs = work.alg_bytes_per_step / (kernel_us * 1e-6) / 1e9
        peak_tf = MFMA_PEAK_TF[args.dtype]
        out = {
            "metric": f"GFLOP/s, batched stride-BRGEMM m=n=k={args.m} {args.dtype}",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "steps_timed": n_timed, "warmup": args.warmup,
            "ms_per_step": round(elapsed / n_timed * 1e3, 5), "timed_region_s": round(elapsed, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"stride-BRGEMM {args.dtype} m=n=k={args.m}, batch={args.batch} independent problems per GPU, br={args.br}, beta={args.beta}"
                                   + (", fused colbias+ReLU" if args.fused else ""),
                       "kernel": headline_kernel, "input_sets_rotated": work.nsets, "per_gpu_batch": args.batch,
                       "streaming_hint": headline_hint},
            "verified": verified, "verify": {"checker": "oracle/liboracle.so (oracle_gemm) on the same inputs", "problems_sampled": vcnt, "normf_rel_max": float(f"{verr:.3g}")},
            "rccl_ranks": world if dist is not None else 0,      # ranks of the process group behind the barrier / MAX reduction (0: a single process without one)
            "pct_mfma_peak": round(100.0 * value / world / 1e3 / peak_tf, 2),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_us": round(kernel_us, 3), "algorithmic_bytes_per_launch": int(work.alg_bytes_per_step),
                         "note": "kernel_us = HIP-event time of the timed region / launches (back-to-back launches from a hipGraph, includes the inter-kernel boundary)"},
            "l3_resident": None if l3_us is None else {"value": round(work.flops_per_step / (l3_us * 1e-6) / 1e9, 1), "unit": "GFLOP/s", "kernel_us": round(l3_us, 3),
                                                       "achieved_GBs": round(work.alg_bytes_per_step / (l3_us * 1e-6) / 1e9, 1)},
            "without_streaming_hint": None if auto_us is None else {"kernel_us": round(auto_us, 3), "frac": round(work.alg_bytes_per_step / (auto_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                                                     "note": "same rotation, libxsmm_hip_set_streaming_hint(0): the library decides -- since round 6 also from the thread's recent operand sets, so an unmodified caller that rotates over more than the Infinity Cache streams like the declared run (round 5: cacheable loads, 10.7 us)"},
            "mfma_busy": busy, "mfma_busy_source": busy_src,
        }
        if l3_us is not None:
            out["l3_resident_us"] = round(l3_us, 3)
        if auto_us is not None:
            out["without_streaming_hint_us"] = round(auto_us, 3)
        if configs:
            out["configs"] = configs
        if round4:
            out["round4"] = round4
        if tpp:
            out["tpp"] = tpp
        if pipelined:
            out["pipelined"] = pipelined
        if roof:
            out["mfma_power_roof_TF"] = roof
            # the rocprofv3 side of the same statement (committed counter passes of the 8192^3 bf16 launch: tools/macro_counters.sh): GRBM_GUI_ACTIVE / kernel time
            # = the clock the chip really ran at, on the drivers' operand values and on zeros, with the matrix-pipe-busy share of both
            try:
                mc = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bf16_macro_counters.json")))[-1]))["counters"]
                pick = lambda d: max((v for k, v in d.items() if "gemm_bf16_macro_kernel<1, 64" in k), key=lambda v: v["kernel_us_under_counters"])     # noqa: E731  (the 8192^3 launch)
                dd, zz = pick(mc["drivers_data"]), pick(mc["zeros"])
                out["effective_clock_GHz"] = {"bf16_m64_blocked_8192": dd["effective_clock_GHz"], "bf16_m64_blocked_8192_on_zeros": zz["effective_clock_GHz"],
                                              "mfma_busy_frac": dd["mfma_busy_frac"], "mfma_busy_frac_on_zeros": zz["mfma_busy_frac"],
                                              "source": "profiles/r05_bf16_macro_counters.json (rocprofv3 --pmc GRBM_GUI_ACTIVE ..., the committed pass; not measured in this process)"}
            except Exception:
                pass
            out["mfma_power_roof_note"] = ("libxsmm_hip_probe_mfma: MFMAs back to back on register operands of the bench's value distribution (no LDS, no memory), "
                                           "one wave per SIMD; MFMA_PEAK_TF is the nominal 2.4 GHz figure, this is what the power budget allows on this data")
        if sweep:
            out["sweep"] = sweep
            out["reuse"] = reuse
            out["ragged"] = ragged
        if sweep:
            measured = {work.label(): (headline_kernel, kernel_us)}
            for grp in (sweep, reuse, ragged, configs, round4, tpp):
                for k, r in grp.items():
                    if isinstance(r, dict) and "us_per_launch" in r:
                        measured[k] = (r.get("kernel", ""), r["us_per_launch"])
            chk = stale_profile_rows(measured)
            if chk is not None:
                out["profiles_check"] = chk
                for msg in chk["stale"]:
                    print(f"bench.py: WARNING: {chk['table']} does not reproduce this run -- {msg} (re-run tools/profile_paths.sh)", file=sys.stderr)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.m, args.dtype, args.br, args.beta, args.fused, args.cpu_seconds, nthreads)
            if sweep:
                out["single_call_us"] = single_call_latency()
                out["sharded_c_abi"] = sharded_c_abi()
            if sweep:       # the reference's classic precision: its f64 JIT kernel for the sweep's f64 32^3 entries, one core (a shorter sample: it is a side figure)
                out["cpu_baseline_f64"] = cpu_baseline(32, "f64", 1, 0, 0, min(args.cpu_seconds, 4.0), 0)
        if args.manifest:
            json.dump({"command": " ".join(sys.argv), "entries": MANIFEST}, open(args.manifest, "w"), indent=1)
    finish(dist, out if rank == 0 else None, args.detail)


def finish(dist, out, detail_path=None):
    """Tear the process group down FIRST, then print the one JSON line as the last thing this process writes to stdout: RCCL announces itself
    on stdout through C stdio ("Librccl path : ..."), whose buffer would otherwise be flushed after Python's at exit and push that text behind
    the JSON line."""
    def flush_all():
        try:
            C.CDLL(None).fflush(None)           # whatever C libraries still hold goes out now ...
        except Exception:
            pass
        sys.stdout.flush(); sys.stderr.flush()
    flush_all()                                 # ... on EVERY rank, before the barrier: the launcher merges the ranks' stdout
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        flush_all()
    if out is not None:
        # the full record first (file + stderr), then the compact line as the LAST thing on stdout
        if detail_path:
            try:
                with open(detail_path, "w") as f:
                    json.dump(out, f, indent=1)
            except OSError as e:
                print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
        sys.stderr.write("bench.py full record: " + json.dumps(out) + "\n"); sys.stderr.flush()
        sys.stdout.write(json.dumps(compact_line(out, detail_path), separators=(",", ":")) + "\n")
        sys.stdout.flush()
    if dist is not None:
        os._exit(0)                             # no exit handler (of the collective library, ...) writes behind the line, on any rank


if __name__ == "__main__":
    main()
