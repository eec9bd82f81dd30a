"""The kernels divide and take square roots with short sequences (a reciprocal and fused corrections:
exact_rcp, exact_div, exact_sqrt in rt_arith.h) where the reference's x86-64 build uses divss and sqrtss
(Source/main.cpp:490-499, 503-514).  They must return the IEEE-754 correctly rounded result, bit for bit.
tools/exact_arith enumerates them on the device against the compiler's `/` and sqrtf: every non-negative
float for the square root, every float of the stated range for the reciprocal, and for the quotient every
dividend significand against a sample of the divisor significands here (every 64th; the whole space of 2^46
pairs takes half a minute and its result is kept in profiles/r03_exact_arith/)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_short_division_and_square_root_are_correctly_rounded():
    exe = os.path.join(ROOT, "tools", "exact_arith")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    out = subprocess.run([exe, "6"], check=True, capture_output=True, text=True, timeout=170).stdout
    checks = {rec["check"]: rec for rec in map(json.loads, out.strip().splitlines())}
    assert set(checks) == {"exact_sqrt", "exact_rcp", "exact_div"}
    for rec in checks.values():
        assert rec["mismatches"] == 0, rec
    assert checks["exact_sqrt"]["inputs"] == 0x7F800000
    assert checks["exact_div"]["pairs"] == (1 << 23) * ((1 << 23) >> 6)


def test_whole_quotient_space_was_enumerated():
    """The committed record of the full run: all 2^46 significand pairs, no mismatch."""
    recs = [json.loads(line) for line in open(os.path.join(ROOT, "profiles", "r03_exact_arith", "exact_arith.json"))]
    full = {r["check"]: r for r in recs}
    assert full["exact_div"]["pairs"] == 1 << 46 and full["exact_div"]["mismatches"] == 0
    assert full["exact_sqrt"]["mismatches"] == 0 and full["exact_rcp"]["mismatches"] == 0
