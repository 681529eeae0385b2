#!/usr/bin/env python
"""Distils gpurun_out/prof_<tag>/ (written by tools/profile_paths.sh on the GPU box) into the small, tracked
files under profiles/:
  <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py` (library kernels only)
  <tag>_bench_under_rocprof.json the bench line printed by that same run
  <tag>_paths_kernel_stats.csv   same for the BASELINE configs #2..#5 (tools/bench_paths.py --headline --eager)
  <tag>_pmc_traffic.json         HBM traffic per launch from the TCC counters: FETCH_SIZE and WRITE_SIZE collected in
                                 separate passes; FETCH_SIZE doubled (gfx950 counts the 128-byte requests of wide
                                 coalesced reads at 64 bytes, MI355X_MICROARCH.md "HBM"); both are reported in KiB.
bench.py reads <tag>_pmc_traffic.json to fill roofline.traffic for the workload it matches."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def ours(name):
    return "xamd::" in name or name.startswith("spmm_jit")


def copy_stats(sub, out):
    files = sorted(glob.glob(os.path.join(src, sub, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if not files:
        return
    rows = list(csv.reader(open(files[-1])))          # the newest run (earlier collections may still lie around)
    with open(os.path.join(dst, out), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if ours(r[0]):
                w.writerow(r)


copy_stats("bench_trace", f"{tag}_bench_kernel_stats.csv")
copy_stats("paths_trace", f"{tag}_paths_kernel_stats.csv")
bj = os.path.join(src, "bench_trace.json")
if os.path.exists(bj) and os.path.getsize(bj):
    open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w").write(open(bj).read())


def counters(sub):
    acc = collections.defaultdict(list)
    files = sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:                              # newest collection only
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].replace(" ", "")].append(float(r["Counter_Value"]))
    return acc


fetch, write = counters("paths_fetch"), counters("paths_write")
out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/profile_paths.sh {tag}",
       "correction": "FETCH_SIZE x2 (gfx950: 128-byte requests of 16-byte-per-lane reads are tallied at 64 bytes); WRITE_SIZE as reported; both KiB",
       "workloads": []}
lines = [json.loads(l) for l in open(os.path.join(src, "paths_fetch.jsonl")) if l.strip().startswith("{")]
for w in lines:
    if "kernel" not in w:
        continue
    key = w["kernel"].replace(" ", "")
    f = [v for k, v in fetch.items() if key in k]
    wr = [v for k, v in write.items() if key in k]
    if not f or not wr:
        continue
    fkb = sum(f[0]) / len(f[0]); wkb = sum(wr[0]) / len(wr[0])
    traffic = int((2 * fkb + wkb) * 1024)
    out["workloads"].append({"workload": w["workload"], "kernel": w["kernel"], "launches_averaged": len(f[0]),
                             "FETCH_SIZE_KiB_avg": round(fkb, 1), "WRITE_SIZE_KiB_avg": round(wkb, 1),
                             "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": w["algorithmic_bytes_per_launch"],
                             "traffic_over_algorithmic": round(traffic / w["algorithmic_bytes_per_launch"], 3)})
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
for w in out["workloads"]:
    print(f"{w['workload'][:80]:80s} {w['kernel'][:34]:34s} traffic/alg = {w['traffic_over_algorithmic']}")
