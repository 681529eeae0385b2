#!/usr/bin/env python3
"""Disassembly of one gfx950 kernel compiled into libxsmm_amd.so (no GPU needed):  python tools/disasm.py <mangled-name-substring> [out.s]"""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import LLVM, ROOT, code_objects  # noqa: E402


def main():
    want = sys.argv[1]
    lib = os.path.join(ROOT, "libxsmm_amd", "lib", "libxsmm_amd.so")
    for image in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(image); f.flush()
            syms = subprocess.check_output([f"{LLVM}/llvm-readelf", "-s", "--wide", f.name], text=True)
            names = [line.split()[-1] for line in syms.splitlines() if want in line and " FUNC " in line]
            if not names:
                continue
            text = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f"--disassemble-symbols={names[0]}", f.name], text=True)
            out = sys.argv[2] if len(sys.argv) > 2 else None
            if out:
                open(out, "w").write(text)
            else:
                sys.stdout.write(text)
            sys.stderr.write(f"{names[0]}: {len(text.splitlines())} lines\n")
            return
    sys.exit(f"no kernel matching {want}")


if __name__ == "__main__":
    main()
