#!/usr/bin/env python
"""DESIGN.md section 5's table of a round, generated from profiles/rNN_bench_detail.json (the full record bench.py writes next to its compact line):
python tools/design_table.py r05 [--write]   -- prints the markdown rows; --write replaces the table that follows the "Round-5 numbers" paragraph of DESIGN.md."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(tag):
    d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_detail.json")))
    out = ["| workload (bench.py group: entry) | kernel | µs per launch | frac of HBM peak | % of MFMA peak |", "|---|---|---|---|---|"]
    r = d["roofline"]
    out.append(f"| **headline** f32 32³ batch 4096, HBM | `{d['config']['kernel']}` | {r['kernel_us']} | **{r['frac']:.3f}** | {d.get('pct_mfma_peak', '–')} |")

    def add(group, key, v):
        if not isinstance(v, dict) or "kernel" not in v:
            return
        us = v.get("us_per_launch", v.get("us"))
        pct = v.get("pct_mfma_peak")
        out.append(f"| {group}: {key} | `{v['kernel']}` | {us} | {v['frac_hbm']:.3f} | {pct if pct is not None else '–'} |")
    for group in ("pipelined", "sweep", "reuse", "ragged", "configs", "round4", "tpp"):
        for key, v in (d.get(group) or {}).items():
            add(group, key, v)
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    table = rows(tag)
    if "--write" not in sys.argv:
        print("\n".join(table)); return
    path = os.path.join(ROOT, "DESIGN.md")
    lines = open(path).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if ln.startswith("| workload (bench.py group: entry) |"))
    end = start
    while end < len(lines) and lines[end].startswith("|"):
        end += 1
    lines[start:end] = table
    open(path, "w").write("\n".join(lines))
    print(f"{len(table) - 2} rows written")


if __name__ == "__main__":
    main()
