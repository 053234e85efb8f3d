#!/usr/bin/env python3
"""Regenerates tests/golden/norm2_flang.json: random vectors and the value amdflang's NORM2
intrinsic returns for them (toolchain behaviour, not reference code).  Needs amdflang."""
import json, os, struct, subprocess, tempfile

here = os.path.dirname(os.path.abspath(__file__))
with tempfile.TemporaryDirectory() as d:
    exe = os.path.join(d, "n2")
    subprocess.check_call(["amdflang", "-O2", os.path.join(here, "make_norm2_vectors.f90"), "-o", exe])
    out = subprocess.check_output([exe]).decode().split("\n")
cases = []
i = 0
while i + 1 < len(out):
    a = out[i].split()
    if len(a) != 2:
        i += 1
        continue
    n = int(a[0])
    xs = out[i + 1].split()
    assert len(xs) == n
    cases.append({"x": xs, "norm2": a[1]})     # hex bit patterns (big-endian IEEE binary64)
    i += 2
cases = cases[:60]      # keep the fixture small
json.dump({"compiler": subprocess.check_output(["amdflang", "--version"]).decode().split("\n")[0],
           "cases": cases}, open(os.path.join(here, "norm2_flang.json"), "w"))
print(len(cases), "cases")
