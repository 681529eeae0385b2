#!/usr/bin/env python
"""Distils gpurun_out/prof_<tag>/ (written by tools/profile_paths.sh on the GPU box) into small files under
gpurun_out/prof_<tag>/summary/ (copied to profiles/ and committed):
  <tag>_bench_kernel_stats.csv   per WORKLOAD of `python bench.py`: rocprofv3 kernel durations of its timed launches.  bench.py's manifest
                                 gives the execution order of the library's launches, so the rotated (HBM-streaming) launches, the
                                 L3-resident leg and every sweep / reuse entry are reported separately -- never mixed in one average.
  <tag>_bench_under_rocprof.json the bench line printed by that same run
  <tag>_pmc_traffic.json         HBM traffic per launch per workload from the TCC counters: FETCH_SIZE and WRITE_SIZE collected in separate
                                 passes; FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes,
                                 MI355X_MICROARCH.md "HBM"); both are reported in KiB.
  <tag>_mfma_busy.json           SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES and GRBM_GUI_ACTIVE per launch per workload, and the derived
                                 matrix-core busy fraction.
  <tag>_copy_floor.csv           kernel durations of tools/headline_probe (copy kernels of the headline footprint, launch-geometry variants)
  <tag>_paths_kernel_stats.csv   rocprofv3 --stats of tools/bench_paths.py --headline (BASELINE configs #2..#5)
bench.py reads <tag>_pmc_traffic.json and <tag>_mfma_busy.json to fill roofline.traffic and mfma_busy for the workloads it matches."""
import collections
import csv
import glob
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)
HBM_PEAK = 8000.0e9
SIMDS = 128        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (8 x the cycles of the launch): 1024 SIMDs / 8


def ours(name):
    if "mfma_probe_kernel" in name:          # libxsmm_hip_probe_mfma: a diagnostic outside the launch manifest
        return False
    return "xamd::" in name or name.startswith("spmm_jit") or name.startswith("pgemm_jit") or name.startswith("meqn_")


def newest(sub, pattern):
    files = sorted(glob.glob(os.path.join(src, sub, "*", pattern)) + glob.glob(os.path.join(src, sub, "*", pattern + ".gz")), key=os.path.getmtime)
    return files[-1] if files else None


def rows_of(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt", newline="") as f:
        yield from csv.DictReader(f)


def library_dispatches(path, value_of):
    """Library launches of a trace / counter file in execution order: list of (kernel name, value)."""
    out = []
    for r in rows_of(path):
        if ours(r["Kernel_Name"]):
            out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], value_of(r)))
    out.sort(key=lambda x: x[0])
    return out


def split_by_manifest(disp, manifest):
    """Walks the dispatches in order and hands each manifest entry its `launches_executed` rows; returns label -> rows of the TIMED launches."""
    res, pos = {}, 0
    for e in manifest["entries"]:
        k = int(e.get("kernels_per_launch", 1))          # a library call may be several kernels (a split chain: partial products + reduction)
        n, nt = e["launches_executed"] * k, e["launches_timed"] * k
        seg = disp[pos:pos + n]
        pos += n
        if len(seg) < n:
            print(f"  !! manifest entry {e['label']}: trace holds {len(seg)} of {n} launches")
        seg = seg[-nt:] if nt <= len(seg) else seg
        if k > 1:                                        # one row per call: the first kernel's name, the values of its k kernels added up
            merged = []
            for i in range(0, len(seg) - k + 1, k):
                grp = seg[i:i + k]
                v0 = grp[0][2]
                val = sum(g[2] for g in grp) if not isinstance(v0, dict) else {c: sum(g[2].get(c, 0.0) for g in grp) for c in set().union(*[g[2].keys() for g in grp])}
                merged.append((grp[0][0], grp[0][1], val))
            seg = merged
        res[e["label"]] = (e, seg)
    if pos != len(disp):
        print(f"  !! {len(disp) - pos} library launches after the last manifest entry")
    return res


trace_us = {}       # label -> rocprofv3 kernel duration of the un-profiled trace pass
# ---- 1. kernel durations per workload -------------------------------------------------------------------------------------
stats_rows = []
kt = newest("bench_trace", "*_kernel_trace.csv")
mf = os.path.join(src, "bench_trace_manifest.json")
if kt and os.path.exists(mf):
    disp = library_dispatches(kt, lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    per = split_by_manifest(disp, json.load(open(mf)))
    with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["workload", "kernel", "timed_launches", "avg_us", "min_us", "max_us", "algorithmic_bytes_per_launch", "GB/s", "frac_hbm", "GFLOP/s", "events_us_in_process"])
        for label, (e, seg) in per.items():
            if not seg:
                continue
            d = [x[2] for x in seg]
            avg = sum(d) / len(d)
            names = collections.Counter(x[1].split("(")[0].replace("void xamd::", "") for x in seg)
            gbs = e["algorithmic_bytes_per_launch"] / (avg * 1e-6)
            trace_us[label] = round(avg, 3)
            w.writerow([label, names.most_common(1)[0][0], len(d), f"{avg:.3f}", f"{min(d):.3f}", f"{max(d):.3f}", e["algorithmic_bytes_per_launch"],
                        f"{gbs / 1e9:.1f}", f"{gbs / HBM_PEAK:.4f}", f"{e['flops_per_launch'] / (avg * 1e-6) / 1e9:.1f}", f"{e['us_per_launch_events']:.3f}"])
            print(f"{label:28s} {names.most_common(1)[0][0][:44]:44s} n={len(d):6d} avg {avg:9.3f} us  frac_hbm {gbs / HBM_PEAK:.3f}")
bj = os.path.join(src, "bench_trace.json")
if os.path.exists(bj) and os.path.getsize(bj):
    open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w").write(open(bj).read())


# ---- 2. HBM traffic -------------------------------------------------------------------------------------------------------------
def counter_pass(sub, names):
    f = newest(sub, "*_counter_collection.csv")
    m = os.path.join(src, sub + "_manifest.json")
    if not f or not os.path.exists(m):
        return {}
    acc = collections.defaultdict(dict)      # dispatch id -> {counter: value}, kernel name
    kn = {}
    for r in rows_of(f):
        if ours(r["Kernel_Name"]) and r["Counter_Name"] in names:
            d = int(r["Dispatch_Id"])
            acc[d][r["Counter_Name"]] = acc[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            acc[d]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            kn[d] = r["Kernel_Name"]
    disp = [(d, kn[d], acc[d]) for d in sorted(acc)]
    return split_by_manifest(disp, json.load(open(m)))


fetch, write = counter_pass("bench_fetch", ["FETCH_SIZE"]), counter_pass("bench_write", ["WRITE_SIZE"])
if fetch and write:
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --eager`, tools/profile_paths.sh {tag}",
           "correction": "FETCH_SIZE x2 (gfx950: 128-byte requests of 16-byte-per-lane reads are tallied at 64 bytes); WRITE_SIZE as reported; both KiB",
           "workloads": []}
    for label, (e, seg) in fetch.items():
        if label not in write or not seg or not write[label][1]:
            continue
        fkb = sum(x[2].get("FETCH_SIZE", 0.0) for x in seg) / len(seg)
        ws = write[label][1]
        wkb = sum(x[2].get("WRITE_SIZE", 0.0) for x in ws) / len(ws)
        traffic = int((2 * fkb + wkb) * 1024)
        out["workloads"].append({"workload": label, "kernel": e["kernel"], "launches_averaged": len(seg), "FETCH_SIZE_KiB_avg": round(fkb, 1), "WRITE_SIZE_KiB_avg": round(wkb, 1),
                                 "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": e["algorithmic_bytes_per_launch"],
                                 "traffic_over_algorithmic": round(traffic / e["algorithmic_bytes_per_launch"], 3)})
        print(f"{label:28s} traffic/alg = {traffic / e['algorithmic_bytes_per_launch']:.3f}")
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)

# ---- 3. matrix-core busy ----------------------------------------------------------------------------------------------------------
mfma = counter_pass("bench_mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"])
if mfma:
    out = {"source": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of `python bench.py --eager`, tools/profile_paths.sh {tag}",
           "definition": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the share of SIMD-cycles of the launch in which the matrix pipe was busy "
                         "(the gfx94x MfmaUtil formula; GRBM_GUI_ACTIVE = cycles the GPU was active for the dispatch, longer than the un-profiled launch because counter collection adds set-up time: "
                         "mfma_busy_frac_unprofiled rescales the same busy cycles to the kernel duration of the trace pass at the clock observed here).  expected_mfma_cycles = the launch's MFMA instructions x their issue "
                         "cycles (f32 32x32x2: 64, 16x16x4: 32; f64 16x16x4: 64; bf16 32x32x16: 32 = 16x16x32: 16 per SIMD; chosen by the kernel's name), a cross-check of the counter's unit.",
           "workloads": {}}
    for label, (e, seg) in mfma.items():
        if not seg:
            continue
        n = len(seg)
        busy = sum(x[2].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for x in seg) / n
        gui = sum(x[2].get("GRBM_GUI_ACTIVE", 0.0) for x in seg) / n
        flops = e["flops_per_launch"]
        # flop per matrix-pipe cycle and SIMD by the KERNEL's MFMA (the manifest's dtype field defaults to f32 for the config workloads: round 3 priced the
        # bf16 BCSC kernel's 16x16x32 MFMAs at the f32 rate and reported counter / expected = 0.062): f32 64, f64 32, bf16 / f16 1024 (32x32x16 in 32 cycles =
        # 16x16x32 in 16), 8-bit 2048
        kname = e.get("kernel", "")
        rate = 32.0 if "f64" in kname else (2048.0 if ("i8" in kname or "fp8" in kname) else (1024.0 if ("bf16" in kname or "f16" in kname or e["dtype"] in ("bf16", "f16")) else 64.0))
        # round 4: the 16^3 bf16 kernels multiply with v_mfma_f32_16x16x16_bf16 (8192 flop in 16 cycles: half the rate of the 32x32x16 / 16x16x32 forms); the integer
        # BCSC kernel with v_mfma_i32_16x16x32_i8 (16 384 operations in 16 cycles) plus one correction MFMA per four (unsigned operand); the ragged f32 kernel works on padded tiles (no cross-check there)
        work = 1.0
        if "bf16_p16" in kname or ("gemm_p16_kernel" in kname and e["dtype"] == "bf16"):
            rate = 512.0
        if "bcsc_mfma_i8" in kname:
            rate, work = 1024.0, 1.25
        if "ragged" in kname:
            work = 0.0                       # the ragged kernel pads and packs problems per tile in shape-specific ways (23^3: 2.02 x the algorithmic MFMA work): no cross-check
        expected = flops * work / rate
        # no cross-check where the launch's MFMA count is not its algorithmic flop count by construction: launches that overlap on several streams (the counters
        # of concurrent kernels cannot be told apart per launch), the masked 8-bit kernel (40^3 problems on 64 x 64 x 32 steps), the bitmask and transform workloads
        if "pipelined" in label or label in ("i8_m40", "u8i8_m40", "bf8_m40", "bitmaskA_8192x64", "vnni2_ld4090"):
            expected = 0.0
        if label == "hf8c8_m64":
            expected = flops / 1024.0            # v_mfma_f32_32x32x16_fp8_fp8: the 16-bit rate per instruction (16 k per step)
        us_pmc = sum(x[2].get("_us", 0.0) for x in seg) / n
        clock_ghz = min(2.4, gui / 8.0 / (us_pmc * 1e3)) if us_pmc > 0 else 0.0   # cycles per ns while the counters were on (GUI_ACTIVE also covers the
                                                                                   # dispatch set-up of a short launch, hence the cap at the 2.4 GHz maximum)
        us_trace = trace_us.get(label)
        out["workloads"][label] = {"kernel": e["kernel"], "launches_averaged": n, "SQ_VALU_MFMA_BUSY_CYCLES": round(busy, 1), "GRBM_GUI_ACTIVE": round(gui, 1),
                                   "kernel_us_with_counters_on": round(us_pmc, 3), "effective_clock_GHz": round(clock_ghz, 3),
                                   "mfma_busy_frac": round(busy / (gui * SIMDS), 4) if gui > 0 else None,
                                   "kernel_us_unprofiled": us_trace,
                                   "mfma_busy_frac_unprofiled": round(busy / (1024.0 * us_trace * 1e3 * clock_ghz), 4) if (us_trace and clock_ghz > 0) else None,
                                   "expected_mfma_cycles": round(expected, 1), "counter_over_expected": round(busy / expected, 3) if expected > 0 else None}
        print(f"{label:28s} mfma busy {busy:14.0f}  gui {gui:10.0f}  frac {busy / (gui * SIMDS) if gui > 0 else 0:.4f}  counter/expected {busy / expected if expected else 0:.3f}")
    json.dump(out, open(os.path.join(dst, f"{tag}_mfma_busy.json"), "w"), indent=1)


# ---- 3b. wave-cycle breakdown / L2 hit rate of selected entries ("sq" mode of profile_paths.sh) -------------------------------------------
SQN = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
sq, l2 = counter_pass("sq_waves", SQN), counter_pass("sq_l2", ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"])
if sq or l2:
    out = {"source": f"rocprofv3 --pmc (SQ pass, TCC pass) of `python bench.py --eager --only ...`, tools/profile_paths.sh {tag} sq",
           "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles summed over waves; WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall (matrix pipe busy, dependency)", "workloads": {}}
    for label in list(sq) + [k for k in l2 if k not in sq]:
        d = {}
        if label in sq and sq[label][1]:
            seg = sq[label][1]
            for c in SQN:
                d[c] = round(sum(x[2].get(c, 0.0) for x in seg) / len(seg), 1)
            wc = d.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            d["share_wait_any"] = round(d.get("SQ_WAIT_ANY", 0.0) / wc, 3); d["share_wait_inst"] = round(d.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3); d["share_active"] = round(d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)
            if d.get("GRBM_GUI_ACTIVE"):
                d["mfma_busy_frac"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] * 128), 4)
        if label in l2 and l2[label][1]:
            seg = l2[label][1]
            for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"):
                d[c] = round(sum(x[2].get(c, 0.0) for x in seg) / len(seg), 1)
            if d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
                d["l2_hit_rate"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
        out["workloads"][label] = d
        print(label, json.dumps(d))
    json.dump(out, open(os.path.join(dst, f"{tag}_wave_breakdown.json"), "w"), indent=1)

# ---- 4./5. plain --stats tables ---------------------------------------------------------------------------------------------------------
def copy_stats(sub, outname, keep):
    f = newest(sub, "*_kernel_stats.csv")
    if not f:
        return
    rows = list(csv.reader(open(f)))
    with open(os.path.join(dst, outname), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(rows[0])
        for r in rows[1:]:
            if keep(r[0]):
                w.writerow(r)


copy_stats("probe_trace", f"{tag}_copy_floor.csv", lambda n: "copy_" in n or "gemm_" in n)
copy_stats("paths_trace", f"{tag}_paths_kernel_stats.csv", ours)
pt = os.path.join(src, "probe_trace.txt")
if os.path.exists(pt):
    open(os.path.join(dst, f"{tag}_headline_probe.txt"), "w").write(open(pt).read())
