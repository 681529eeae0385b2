#!/usr/bin/env python3
"""Register / LDS / scratch budget of every gfx950 kernel compiled into libxsmm_amd.so, read from the code objects' own metadata
(no GPU needed).  The library embeds one clang offload bundle per .hip translation unit in its .hip_fatbin section; each bundle's
gfx950 entry is an ELF whose AMDGPU note lists, per kernel, the VGPR / AGPR / SGPR counts, the static LDS size and the scratch
(private segment) size.  Scratch > 0 on a hot kernel means spills or a by-reference kernel-argument block: the first thing to fix.

  python tools/kernel_resources.py                 # table on stdout
  python tools/kernel_resources.py --out profiles/r02_kernel_resources.txt
"""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """Yields the gfx950 ELF images embedded in `lib`."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
    at = blob.find(MAGIC)
    while at >= 0:
        (count,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
        cursor = at + len(MAGIC) + 8
        for _ in range(count):
            offset, size, triple_len = struct.unpack_from("<QQQ", blob, cursor)
            triple = blob[cursor + 24:cursor + 24 + triple_len].decode()
            cursor += 24 + triple_len
            if "gfx950" in triple and size:
                yield blob[at + offset:at + offset + size]
        at = blob.find(MAGIC, at + len(MAGIC))


FIELDS = (".name", ".vgpr_count", ".agpr_count", ".sgpr_count", ".group_segment_fixed_size", ".private_segment_fixed_size",
          ".max_flat_workgroup_size", ".vgpr_spill_count", ".sgpr_spill_count")


def kernels_of(image):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", f.name], text=True)
    rows, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*(?:- )?(\.[a-z_]+):\s*(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip().strip("'\"")
        if key == ".agpr_count" and (cur is None or ".agpr_count" in cur):   # first field of a kernel record (keys are sorted)
            cur = {}
            rows.append(cur)
        if cur is not None and key in FIELDS:
            cur[key] = val
    return [r for r in rows if ".name" in r]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True, check=True).stdout
    return out.splitlines()


def waves_per_simd(total_regs):
    """gfx950: 512 registers per SIMD lane shared by VGPRs and AGPRs, allocated in blocks of 8; at most 8 waves per SIMD.
    The metadata's .vgpr_count is the unified total (architectural + accumulation registers)."""
    total = -(-max(total_regs, 1) // 8) * 8
    return min(8, 512 // total)


def waves_per_simd_lds(lds_bytes, wg_size):
    """Workgroups of one CU share 160 KiB of LDS; a workgroup's waves spread over the CU's four SIMDs."""
    if lds_bytes <= 0:
        return 8
    return min(8, max(1, (160 * 1024 // lds_bytes) * max(wg_size // 64, 1) // 4))


def collect(lib):
    """`lib`: a shared library with embedded offload bundles, or a bare gfx950 code object (what hiprtc returns: LIBXSMM_HIP_JIT_DUMP)."""
    rows = []
    head = open(lib, "rb").read(64)
    images = [open(lib, "rb").read()] if (lib.endswith(".co") and head[:4] == b"\x7fELF" and MAGIC not in head) else code_objects(lib)
    for image in images:
        rows.extend(kernels_of(image))
    pretty = demangle([r[".name"] for r in rows])
    table = []
    for r, name in zip(rows, pretty):
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*\)$", "", name)
        v, a = int(r.get(".vgpr_count", 0)), int(r.get(".agpr_count", 0))
        table.append(dict(name=name, vgpr=v, agpr=a, sgpr=int(r.get(".sgpr_count", 0)), lds=int(r.get(".group_segment_fixed_size", 0)),
                          scratch=int(r.get(".private_segment_fixed_size", 0)), spills=int(r.get(".vgpr_spill_count", 0)),
                          wg=int(r.get(".max_flat_workgroup_size", 0)), waves=0))
        table[-1]['waves'] = min(waves_per_simd(v), waves_per_simd_lds(table[-1]['lds'], table[-1]['wg']))
    table.sort(key=lambda t: t["name"])
    return table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "libxsmm_amd", "lib", "libxsmm_amd.so"))
    ap.add_argument("--out")
    args = ap.parse_args()
    table = collect(args.lib)
    lines = [f"# {len(table)} gfx950 kernels in {os.path.relpath(args.lib, ROOT)} (code-object metadata; regs = VGPRs + AGPRs as allocated, agpr = the AGPR share; waves/SIMD = the smaller of the register and the LDS limit, workgroup size = the launch bound)",
             f"{'regs':>5} {'agpr':>5} {'sgpr':>5} {'lds_B':>7} {'scratch_B':>9} {'spills':>6} {'wg':>5} {'waves/SIMD':>10}  kernel"]
    for t in table:
        lines.append(f"{t['vgpr']:5d} {t['agpr']:5d} {t['sgpr']:5d} {t['lds']:7d} {t['scratch']:9d} {t['spills']:6d} {t['wg']:5d} {t['waves']:10d}  {t['name']}")
    text = "\n".join(lines) + "\n"
    if args.out:
        open(args.out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
