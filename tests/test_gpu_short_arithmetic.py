"""rt_math.h replaces the compiler's correctly rounded 1/x, sqrt and a/b by sequences that cost about half -- allowed only because they
return the same bits.  That claim is not sampled, it is enumerated: tools/verify/short_arithmetic.hip includes the product's headers, is
built with the product's flags, and compares the shipped functions with the compiler's forms over their whole argument spaces (all 2^32
floats; every non-negative float; 8 x 2^17 x 2^23 pairs of significands for the division -- the full 2^46 take 43 s and are logged in
profiles/r03_microbench_division.txt).  Bit parity with the reference (vec3.rs:37-44, cube.rs:72-74, quad.rs:91, renderer.rs:96-97) rests on it."""
import subprocess

import pytest

from conftest import pkg


@pytest.mark.gpu
def test_short_reciprocal_sqrt_division_equal_the_correctly_rounded_forms():
    exe = pkg("build").build_verify()
    r = subprocess.run([exe, "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    out = r.stdout
    assert "recip_normal_range: 0 differences inside the proven range" in out and "46137340 outside it" in out
    assert "recip3<true>: 0 differences over all 2^32 x" in out
    assert "length_for_normalize: 0 contract violations" in out
    assert "div_bounded, div_by_rn: 0 differences over 8 x 2^17" in out and "div_bounded: 0 differences on zeros" in out
    assert "u32_to_range11: 0 differences over all 2^23 mantissas" in out and "all checks passed" in out
