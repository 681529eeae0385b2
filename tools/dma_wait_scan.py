"""Which kernels have an `s_waitcnt vmcnt(0)` directly in front of an LDS read / write while they use LDS-DMA (global_load_lds / buffer_load ... lds)?
The compiler puts that wait in front of every LDS access it can see while LDS-DMA requests may be in flight (it cannot tell the accessed bytes from the
requests' destination): inside a loop that keeps a ring of requests in flight it lands the whole ring before every step.  Disassembles the gfx950 code
objects of the built library (no GPU needed): python tools/dma_wait_scan.py [lib]   -> kernel, DMA instructions, suspicious waits (and the instruction behind each)."""
import os, re, subprocess, sys, tempfile, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libxsmm_amd", "lib", "libxsmm_amd.so")
for image in kr.code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image); f.flush()
        dis = subprocess.check_output([f"{kr.LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f.name], text=True)
    name, body = None, {}
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1); body[name] = []; continue
        if name and line.strip():
            body[name].append(line.strip().split("//")[0].strip())
    for name, ins in body.items():
        dma = sum(1 for x in ins if x.startswith("global_load_lds") or (x.startswith("buffer_load") and " lds" in x))
        if not dma:
            continue
        sus = collections.Counter()
        for i, x in enumerate(ins[:-1]):
            if re.match(r"s_waitcnt vmcnt\(0\)$", x) and re.match(r"ds_(read|write)", ins[i + 1]):
                sus[ins[i + 1].split()[0]] += 1
        dem = subprocess.run(["c++filt", name], text=True, capture_output=True).stdout.strip().split("(")[0]
        print(f"{dem[:110]:110s} dma {dma:3d}  vmcnt(0)+LDS access: {dict(sus) if sus else '-'}")
