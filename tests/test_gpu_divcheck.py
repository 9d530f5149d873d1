"""STRICT arithmetic divides by the hardware's own reciprocal / Newton / correction sequence WITHOUT the compiler's
v_div_scale / v_div_fmas / v_div_fixup frame (lsf_cell.hpp: recip_refined, div_by).  That is only legitimate if it returns
the bits of `n / d` wherever the library uses it: brute force over random operands (divisor exponents up to +-660, the
range of the WENO divisors) on the device, besides the reference fixtures every STRICT parity test compares with.  The
same for the square roots of the cell update (lsf_cell.hpp: sqrt_unframed against the compiler's sqrt)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_unframed_division_returns_the_bits_of_the_ieee_division(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not on this box")
    exe = tmp_path / "divcheck"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-I", os.path.join(ROOT, "levelsetfortran_amd", "csrc"), "-o", str(exe),
                    os.path.join(ROOT, "profiles", "micro", "divcheck.hip")], check=True, timeout=600)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=600).stdout
    rows = re.findall(r"divisor exponents \+-(\d+): (\S+) quotients; mismatches vs n / d:  A \(refined reciprocal, 1 correction\) (\d+)"
                      r".*unframed sqrt vs sqrt (\d+)", out)
    assert [r[0] for r in rows] == ["6", "660"], out
    for _, count, bad, bad_sqrt in rows:
        assert float(count) > 8e9 and int(bad) == 0 and int(bad_sqrt) == 0, out
