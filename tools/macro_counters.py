"""Distils tools/macro_counters.sh: per blocked-GEMM kernel and operand data (the drivers' values / zeros) the counters of the timed launches ->
profiles/<tag>_bf16_macro_counters.{txt,json}.  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs: cycles of the launch = GUI / 8, effective clock = cycles / kernel time
(MI355X_MICROARCH.md, "DVFS give-back"); SQ_VALU_MFMA_BUSY_CYCLES / (GUI / 8 x 1024 SIMDs) = share of SIMD-cycles with the matrix pipe busy; the SQ wave counters are
quad-cycles summed over waves (shares of SQ_WAVE_CYCLES: ACTIVE_INST_ANY issuing, WAIT_INST_ANY issue stall -- matrix pipe busy / dependency --, WAIT_ANY parked)."""
import csv, glob, gzip, json, os, sys, collections

src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_MFMA"]
FLOP = {"bf16_m64_blocked": 2.0 * 4096 ** 3, "bf16_m64_blocked_8192": 2.0 * 8192 ** 3, "bf16_m32_blocked": 2.0 * 4096 ** 3, "f32_m64_blocked": 2.0 * 4096 ** 3}


def rows(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt", newline="") as f:
        yield from csv.DictReader(f)


def one_pass(sub):
    files = sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv*")), key=os.path.getmtime)
    if not files:
        return {}
    acc, kn = collections.defaultdict(dict), {}
    for r in rows(files[-1]):
        n = r["Kernel_Name"]
        if "macro_kernel" not in n and "blocked_kernel" not in n:
            continue
        d = int(r["Dispatch_Id"])
        acc[d][r["Counter_Name"]] = acc[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        acc[d]["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        kn[d] = n.split("(")[0].replace("void xamd::", "") + f" grid={r['Grid_Size']}"      # (4096^3 and 8192^3 out of 64^3 tiles share a kernel: the grid tells them apart)
    by_kernel = collections.defaultdict(list)
    for d in sorted(acc):
        by_kernel[kn[d]].append(acc[d])
    out = {}
    for k, ds in by_kernel.items():
        ds = ds[len(ds) // 2:]                      # the later half: warm
        m = {c: sum(x.get(c, 0.0) for x in ds) / len(ds) for c in NAMES + ["us"]}
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        wc = m["SQ_WAVE_CYCLES"] or 1.0
        out[k] = {"launches_averaged": len(ds), "kernel_us_under_counters": round(m["us"], 2), "GRBM_GUI_ACTIVE": round(m["GRBM_GUI_ACTIVE"], 0),
                  "effective_clock_GHz": round(cyc / (m["us"] * 1e3), 3) if m["us"] else None,
                  "mfma_busy_frac": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4) if cyc else None,
                  "share_active_inst": round(m["SQ_ACTIVE_INST_ANY"] / wc, 3), "share_wait_inst": round(m["SQ_WAIT_INST_ANY"] / wc, 3), "share_wait_any": round(m["SQ_WAIT_ANY"] / wc, 3),
                  "SQ_VALU_MFMA_BUSY_CYCLES": round(m["SQ_VALU_MFMA_BUSY_CYCLES"], 0), "SQ_WAVE_CYCLES": round(m["SQ_WAVE_CYCLES"], 0), "SQ_INSTS_MFMA": round(m["SQ_INSTS_MFMA"], 0)}
    return out


def unprofiled(name):
    p = os.path.join(src, name)
    try:
        last = [l for l in open(p).read().splitlines() if l.startswith("{")][-1]
        return {k: {"us": v["us_per_launch"], "TFLOP/s": round(v["GFLOP/s"] / 1e3, 1), "kernel": v["kernel"], "verified": v.get("verified")} for k, v in json.loads(last)["results"].items()}
    except Exception as e:
        return {"error": repr(e)}


res = {"source": "tools/macro_counters.sh (rocprofv3 --pmc ... --kernel-trace, eager launches) + un-profiled hipGraph timing of the same entries",
       "counters": {"drivers_data": one_pass("data"), "zeros": one_pass("zero")},
       "unprofiled": {"drivers_data": unprofiled("unprofiled_data.json"), "zeros": unprofiled("unprofiled_zero.json")}}
for data, ks in res["counters"].items():
    for k, v in ks.items():
        print(f"{data:13s} {k[:64]:64s} us {v['kernel_us_under_counters']:9.2f}  clock {v['effective_clock_GHz']} GHz  mfma busy {v['mfma_busy_frac']}  active {v['share_active_inst']} wait_inst {v['share_wait_inst']} wait_any {v['share_wait_any']}")
for data, ks in res["unprofiled"].items():
    print(data, json.dumps(ks))
tag = os.path.basename(os.path.normpath(src)).replace("macro_", "") or "r06"      # gpurun_out/macro_<tag>
json.dump(res, open(os.path.join(src, f"{tag}_bf16_macro_counters.json"), "w"), indent=1)
