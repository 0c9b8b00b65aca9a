"""The C ABI used from plain C (no Python, no torch in the process): tests/c/abi_smoke.c."""
import os
import subprocess

import numpy as np
import pytest

from oracle import cfmm_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_plain_c_client(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "cfmmrouter.jl_amd")
    subprocess.run(["gcc", "-O1", "-std=c11", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L", libdir, "-lcfmm_amd",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
    print(out)
    D, L = orc.product_find_arb([1e6, 1e6], 1.0, [2.0, 1.0])
    assert D[1] > 0 and L[0] > 0
    line = [l for l in out.splitlines() if l.startswith("trades pool1")][0]
    nums = [float(x) for x in line.split(":", 1)[1].replace("D=", " ").replace("L=", " ").replace("[", " ")
            .replace("]", " ").replace(",", " ").split()]
    assert nums == [D[0], D[1], L[0], L[1]]          # %.17g round-trips binary64: bit-exact through the C client
    assert "route: v=[1.0008" in out and "171.40" in out      # exit code 0 already checked Ψ ≈ [0, 171.4]
    assert "two token indices must differ" in out


def test_plain_c_client_compiles():
    """CPU: the header is valid C11 and the client links against the library."""
    libdir = os.path.join(ROOT, "cfmmrouter.jl_amd")
    subprocess.run(["gcc", "-O1", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", "/tmp/abi_smoke_cpu", "-L", libdir,
                    "-lcfmm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"], check=True)
    r = subprocess.run(["/tmp/abi_smoke_cpu"], capture_output=True, text=True, timeout=120)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode == 2 and "no CPU fallback" in r.stderr      # fails loudly without a device
