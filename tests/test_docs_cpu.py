"""The documents a maintainer reads against the sources they describe (no GPU, no library needed).

INTEGRATION.md 3a lists every environment switch the library reads: a switch added to the sources without a line there (or a documented one that no longer
exists) fails here.  include/*.h cite the reference interface they replace (file:line) -- the boundary rule of this repo."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libxsmm_amd", "csrc")


def _env_switches_in_sources():
    names = set()
    for path in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(CSRC, "*.hpp")):
        names |= set(re.findall(r'getenv\("(LIBXSMM_HIP_[A-Z0-9_]+)"\)', open(path).read()))
    return names


def _env_switches_documented():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    start = text.index("## 3a. Environment switches")
    section = text[start:text.index("\n## ", start + 10)]
    names = set()
    for cell in re.findall(r"`([^`]+)`", section):
        full = None                                            # `LIBXSMM_HIP_F32_LEAN`, `_F32_BLOB`, ...: a short form continues the last full name's prefix
        for token in re.split(r"[ ,/]+", cell):
            token = re.sub(r"=.*$", "", token)
            if re.fullmatch(r"LIBXSMM_HIP_[A-Z0-9_]+", token):
                full = token
                names.add(token)
            elif re.fullmatch(r"_[A-Z0-9_]+", token):
                names.add("LIBXSMM_HIP" + token)
    return names


def test_every_environment_switch_is_documented():
    src, doc = _env_switches_in_sources(), _env_switches_documented()
    assert 10 < len(src) < 20, sorted(src)            # round 6: the settled A/B switches of rounds 2-5 are constants now (review item: < 20 documented switches)
    assert not (src - doc), f"read by the library but missing from INTEGRATION.md 3a: {sorted(src - doc)}"
    assert not (doc - src), f"documented in INTEGRATION.md 3a but read nowhere: {sorted(doc - src)}"


def test_public_headers_cite_the_reference_interface():
    for header in ("libxsmm.h", "libxsmm_hip.h", "libxsmm_utils.h"):
        path = os.path.join(ROOT, "include", header)
        assert os.path.exists(path), header
        text = open(path).read()
        cites = re.findall(r"(?:src|include|samples)/[A-Za-z0-9_./]+\.(?:c|h|cpp):\d+", text)
        assert len(cites) >= 3, f"{header}: {len(cites)} reference citations (file:line)"


def test_design_section5_table_is_the_committed_bench_record():
    """DESIGN.md's closing-run table is generated (tools/design_table.py) from profiles/r06_bench_detail.json: every row of the generator's output stands in the document, so
    the numbers a reader sees are the ones of the committed closing run."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import design_table
    doc = open(os.path.join(ROOT, "DESIGN.md")).read()
    rows = design_table.rows("r06")
    assert len(rows) > 70
    missing = [r for r in rows if r not in doc]
    assert not missing, missing[:3]
