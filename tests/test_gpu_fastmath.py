"""The arithmetic claim behind option "fast_math": rcp_refined / div_by / fast_sqrt of sweep_kernels.hip return the
bits of the compiler's IEEE / and sqrt for every operand inside the window -- tests/native/fastmath_check.hip, a
stand-alone HIP program (2^30 random operand pairs on the MI355X)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "fastmath_check.hip")


def _build(exe):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", SRC, "-o", exe], check=True)


def test_fastmath_check_compiles(tmp_path):
    _build(str(tmp_path / "fastmath_check"))


@pytest.mark.gpu
def test_fast_division_and_sqrt_match_the_ieee_sequences(tmp_path):
    exe = str(tmp_path / "fastmath_check")
    _build(exe)
    r = subprocess.run([exe, "1024"], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0
    line = [l for l in r.stdout.splitlines() if l.startswith("FASTMATH_CHECK")][0]
    f = dict(kv.split("=") for kv in line.split()[1:])
    assert int(f["pairs"]) == 4096 * 256 * 1024
    assert f["div_mismatch"] == "0" and f["sqrt_mismatch"] == "0" and f["zero_mismatch"] == "0"
    # fast_exp (log-space GeometricMean form): within an ulp of the device library's exp almost everywhere, never more
    # than 2 apart, and < 1 ulp from the long-double exponential on the sample the host re-computes
    assert int(f["exp_max_ulp_vs_lib"]) <= 2
    assert int(f["exp_over_1ulp_vs_lib"]) <= int(f["pairs"]) // 1000
    assert float(f["exp_max_err_ulp_vs_expl"]) < 1.0
