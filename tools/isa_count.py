#!/usr/bin/env python3
"""Instruction mix of the hot loops of a kernel: compiles imdgpu.hip (or the file given) to gfx950 assembly and prints, for every
basic block of the named kernels with at least MIN LDS reads, the instruction counts by class.  Runs in the dev container (no GPU).
usage: tools/isa_count.py <mangled-name-substring> [...] [--file imd_amd/csrc/imdgpu.hip] [--min-lds 6] [--extra "-D..."]"""
import argparse, os, re, subprocess, sys, tempfile
ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--file", default="imd_amd/csrc/imdgpu.hip")
ap.add_argument("--min-lds", type=int, default=6)
ap.add_argument("--extra", default="")
ap.add_argument("--asm", default=None, help="reuse / keep the assembly file")
a = ap.parse_args()
asm = a.asm or os.path.join(tempfile.gettempdir(), "isa_count.s")
if not (a.asm and os.path.exists(asm)):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                           "--cuda-device-only", "-S", "-o", asm, a.file] + a.extra.split(), stderr=subprocess.DEVNULL)
s = open(asm).read()
F64 = re.compile(r"v_(fma|add|mul|max|min|fract|cvt_i32|cvt_u32|cmp\w*|cndmask\w*|div\w*|rcp|floor|trunc|ldexp)_?\w*f64")
for m in re.finditer(r"^(_Z\w+):[^\n]*\n", s, re.M):
    nm = m.group(1)
    if not any(x in nm for x in a.names):
        continue
    body = s[m.end(): s.index(".Lfunc_end", m.end())]
    vg = re.search(r"\.vgpr_count:\s+(\d+)", s[s.find(nm, s.find("amdhsa.kernels")):] or "")
    sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", s[s.find(nm, s.find("amdhsa.kernels")):] or "")
    print("%s  vgpr %s spill %s scratch-ops %d" % (nm, vg.group(1) if vg else "?", sp.group(1) if sp else "?", body.count("scratch_")))
    cur = None
    for l in body.split("\n"):
        b = re.match(r"^(\.LBB\d+_\d+):", l)
        if b:
            if cur and cur["lds"] >= a.min_lds:
                print("   %-14s insts %4d valu %4d (f64 %3d) lds %3d salu %3d vmem %2d scratch %d waitcnt %d" %
                      (cur["n"], cur["i"], cur["v"], cur["f"], cur["lds"], cur["s"], cur["g"], cur["x"], cur["w"]))
            cur = dict(n=b.group(1), i=0, v=0, f=0, lds=0, s=0, g=0, x=0, w=0)
        elif cur is not None and l.startswith("\t") and not l.strip().startswith((".", ";")):
            t = l.strip()
            cur["i"] += 1
            if t.startswith("v_"): cur["v"] += 1
            if "f64" in t.split()[0]: cur["f"] += 1
            if t.startswith("ds_"): cur["lds"] += 1
            if t.startswith("s_") and not t.startswith("s_waitcnt"): cur["s"] += 1
            if t.startswith(("global_", "buffer_", "flat_")): cur["g"] += 1
            if t.startswith("scratch_"): cur["x"] += 1
            if t.startswith("s_waitcnt"): cur["w"] += 1
    if cur and cur["lds"] >= a.min_lds:
        print("   %-14s insts %4d valu %4d (f64 %3d) lds %3d salu %3d vmem %2d scratch %d waitcnt %d" %
              (cur["n"], cur["i"], cur["v"], cur["f"], cur["lds"], cur["s"], cur["g"], cur["x"], cur["w"]))
