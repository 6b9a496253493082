"""CPU-side checks: the C-ABI library builds for gfx950, loads, and exports every
symbol include/imdgpu.h declares (no compute calls without a GPU)."""
import os
import re

import imd_amd
from imd_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    lib.build()
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "imdgpu.h")).read()
    hdr += open(os.path.join(ROOT, "include", "imdgpu_ttm.h")).read() if os.path.exists(
        os.path.join(ROOT, "include", "imdgpu_ttm.h")) else ""
    declared = set(re.findall(r"\b(imdgpu_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, "header symbols not exported: %s" % missing
    assert set(lib.SYMBOLS) <= declared


def test_product_does_not_import_oracle():
    """The shipped package must not import, include or load anything under oracle/."""
    pkg = os.path.join(ROOT, "imd_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle|#\s*include\s*[<\"][^>\"]*oracle|liborcimd|pyoracle", re.M)
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                assert not pat.search(txt), "%s references the oracle" % os.path.join(dp, f)
