"""The sample loop's u and v (Source/main.cpp:627-628) are double quotients by the frame width and
height.  The kernels form them from a host-rounded reciprocal with two fused multiply-adds
(divide_by_reciprocal in rt_arith.h); tests/native/div_check.c repeats that sequence on the host
and compares it with the division, bit for bit, over every width up to 4096, random widths up to
2^24 and random pixel + jitter numerators."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_quotient_from_rounded_reciprocal_is_the_correctly_rounded_quotient(tmp_path):
    exe = str(tmp_path / "div_check")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(HERE, "native", "div_check.c"), "-lm"])
    bad, n = (int(v) for v in subprocess.check_output([exe, "600"]).split())
    assert n > 10_000_000 and bad == 0
