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


def _build_c(name, tmp_path):
    exe = str(tmp_path / name)
    libdir = os.path.join(ROOT, "cfmmrouter.jl_amd")
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", name + ".c"), "-o", exe, "-L", libdir, "-lcfmm_amd",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"], check=True)
    return exe


@pytest.mark.gpu
def test_rccl_all_reduce_from_a_plain_c_host(tmp_path):
    """north_star's collective -- "RCCL all-reduce of psi and grad g over xGMI per outer iteration" -- at the C ABI, from a
    process with no Python and no torch in it: cfmm_rccl_unique_id + cfmm_rccl_init_rank (world 1 on the 1-GPU box: the
    communicator path end to end; tests/c/abi_rccl.c N forks N ranks on an N-GPU node), the global psi through
    ncclAllReduce on the context's stream, route! on the sharded context, cfmm_set_rccl_comm(NULL)."""
    exe = _build_c("abi_rccl", tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0
    assert "rank 0/1" in r.stdout and "route!" in r.stdout


@pytest.mark.gpu
@pytest.mark.xfail(strict=False, reason="first contact with a multi-GPU node: this path has never crossed a physical xGMI link; a "
                                        "failure is reported (XFAIL), a pass too (XPASS), neither stops the suite")
def test_rccl_all_reduce_across_physical_gpus_if_there_are_several(tmp_path):
    """The same plain-C client with one rank per GPU (processes forked before any HIP call, the 128-byte id handed over through pipes): every
    rank holds 1 / N of the market and must return the psi of the WHOLE market -- north_star's collective over xGMI.  Needs
    N >= 2 GPUs: skipped on the 1-GPU box, runs wherever the suite meets a multi-GPU node."""
    import torch
    n_gpus = torch.cuda.device_count()
    if n_gpus < 2:
        pytest.skip("one GPU visible: the multi-rank RCCL run needs at least two")
    world = min(n_gpus, 8)
    exe = _build_c("abi_rccl", tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, str(world)], capture_output=True, text=True, timeout=600, env=env)
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0 and "RCCL_ABI_OK" in r.stdout
    assert r.stdout.count("rel err vs unsharded") == world


def test_rccl_client_compiles():
    """CPU: the RCCL client is valid C against the header and links against the library (no RCCL needed to link: it is
    resolved at first use)."""
    exe = _build_c("abi_rccl", __import__("pathlib").Path("/tmp"))
    r = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=120)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode == 2 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_rccl_exchange_through_the_python_binding():
    """The same entry points through ctypes, world 1: eval / find_arb / sweep_dev / route on a context with an RCCL
    communicator return what the plain context returns (one rank: the all-reduce is the identity), pre-arming is off,
    cfmm_set_peers is refused while the communicator is active, and the exchange can be switched off again."""
    import torch
    import cfmmrouter_amd as cr
    from cfmmrouter_amd import synth
    n = 48
    batches = [synth.product_pools(60_000, n, seed=91), synth.geomean_pools(20_000, n, seed=92)]
    v = synth.sweep_prices(n, seed=93)
    plain = cr.DeviceBackend(n, batches)
    be = cr.DeviceBackend(n, batches)
    try:
        psi0, acc0 = plain.eval(v)
        be.ctx.rccl_init_rank(be.ctx.rccl_unique_id(), 1, 0)
        psi1, acc1 = be.eval(v)
        np.testing.assert_array_equal(psi1, psi0)
        assert acc1 == acc0
        psi2, _ = be.find_arb(v)
        np.testing.assert_array_equal(psi2, plain.find_arb(v)[0])
        for a, b in zip(be.trades(), plain.trades()):
            np.testing.assert_array_equal(a, b)
        vt = torch.from_numpy(v).cuda()
        ot = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
        be.ctx.sweep_dev(vt.data_ptr(), ot.data_ptr(), False)
        torch.cuda.synchronize()
        np.testing.assert_allclose(ot.cpu().numpy()[:n], psi0, rtol=0, atol=1e-14 * np.max(np.abs(psi0)))
        with pytest.raises(RuntimeError, match="one exchange at a time"):
            be.ctx.set_peers([0], 1, 0, 0)
        obj = cr.LinearNonnegative(synth.linear_prices(n, seed=94))
        r1 = cr.Router(obj, batches, n, _backend=be)
        r0 = cr.Router(obj, batches, n, _backend=plain)
        cr.route_(r1, v=np.ones(n), solver="native")
        cr.route_(r0, v=np.ones(n), solver="native")
        np.testing.assert_array_equal(r1.v, r0.v)                  # launch-when-ready vs pre-armed: the same bits
        np.testing.assert_array_equal(cr.netflows(r1), cr.netflows(r0))
        be.ctx.set_rccl_comm(None)
        np.testing.assert_array_equal(be.eval(v)[0], psi0)
    finally:
        be.close()
        plain.close()
