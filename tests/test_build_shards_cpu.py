"""gemm_kernels.hip is compiled as one main translation unit (host half, every kernel template instantiation `extern`) plus N shard units that each generate the code
of their share of the instantiations (libxsmm_amd/csrc/gemm_shards/, written by tools/gen_gemm_shards.py; round-4 review item 9: the fresh build went from 4 min 32 s to
under a minute on eight cores).  A missing instantiation cannot go unnoticed -- it is an undefined symbol at link time -- but the lists can rot in the other direction
(instantiations nothing launches any more) and the shards can drift out of balance; this pins the bookkeeping against the BUILT library."""
import glob
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SH = os.path.join(ROOT, "libxsmm_amd", "csrc", "gemm_shards")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _decls(path):
    out = []
    for line in open(path):
        m = re.match(r"^(extern )?template __global__ void (\w+<.*>)\((.*)\);$", line.strip())
        if m:
            out.append(m.group(2))
    return out


def test_every_extern_instantiation_lives_in_exactly_one_shard():
    ext = _decls(os.path.join(SH, "extern.inc"))
    shards = [_decls(p) for p in sorted(glob.glob(os.path.join(SH, "shard_*.inc")))]
    assert len(shards) >= 4 and len(ext) > 250
    flat = [d for s in shards for d in s]
    assert sorted(flat) == sorted(ext) and len(set(flat)) == len(flat)
    sizes = [len(s) for s in shards]
    assert max(sizes) - min(sizes) <= 1, sizes                        # dealt round-robin: balanced by construction
    mk = open(os.path.join(ROOT, "libxsmm_amd", "csrc", "Makefile")).read()
    assert f"NSHARDS := {len(shards)}" in mk


def test_the_built_library_holds_exactly_the_listed_instantiations():
    import kernel_resources as kr
    lib = os.path.join(ROOT, "libxsmm_amd", "lib", "libxsmm_amd.so")
    if not (os.path.exists(lib) and os.path.exists(os.path.join(kr.LLVM, "llvm-readelf"))):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    import gen_gemm_shards as gen
    templates = gen.sharded_templates()
    built = set()
    for t in kr.collect(lib):
        m = re.match(r"^xamd::(\w+)(<.*>)$", t["name"])
        if m and m.group(1) in templates:
            built.add(m.group(1) + m.group(2))
    listed = set(_decls(os.path.join(SH, "extern.inc")))
    assert built == listed, (sorted(built - listed)[:5], sorted(listed - built)[:5])
