"""Which kernels have an `s_waitcnt vmcnt(0)` directly in front of an LDS read / write while they use LDS-DMA (global_load_lds / buffer_load ... lds)?
The compiler puts that wait in front of every LDS access it can see while LDS-DMA requests may be in flight (it cannot tell the accessed bytes from the
requests' destination): inside a loop that keeps a ring of requests in flight it lands the whole ring before every step (decision 36).  A single such
wait can also be a kernel's own (one problem per wave, nothing else in flight).  Disassembles the gfx950 code objects of the built library (no GPU needed):
python tools/dma_wait_scan.py [lib]   -> kernel, DMA instructions, suspicious waits (and the instruction behind each).  scan(lib) returns the same as a dict;
instructions(lib, mangled_substring) the instruction list of the first kernel whose mangled name contains the substring."""
import collections, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr

DEFAULT_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libxsmm_amd", "lib", "libxsmm_amd.so")


def bodies(lib):
    """Yields (mangled name, [instructions]) of every kernel / function in the library's gfx950 code objects."""
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
        yield from body.items()


def scan(lib=DEFAULT_LIB):
    """{demangled kernel name: (LDS-DMA instructions, {LDS instruction directly behind an s_waitcnt vmcnt(0): count})} for the kernels that use LDS-DMA."""
    out = {}
    for name, ins in bodies(lib):
        dma = sum(1 for x in ins if x.startswith("global_load_lds") or (x.startswith("buffer_load") and " lds" in x))
        if not dma:
            continue
        sus = collections.Counter()
        for i, x in enumerate(ins[:-1]):
            if re.match(r"s_waitcnt vmcnt\(0\)$", x) and re.match(r"ds_(read|write)", ins[i + 1]):
                sus[ins[i + 1].split()[0]] += 1
        dem = subprocess.run(["c++filt", name], text=True, capture_output=True).stdout.strip().split("(")[0]
        out[dem] = (dma, dict(sus))
    return out


def instructions(lib, mangled_substring):
    for name, ins in bodies(lib):
        if mangled_substring in name:
            return ins
    return None


if __name__ == "__main__":
    for dem, (dma, sus) in sorted(scan(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB).items()):
        print(f"{dem[:110]:110s} dma {dma:3d}  vmcnt(0)+LDS access: {sus if sus else '-'}")
