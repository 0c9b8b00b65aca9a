"""Register / LDS / scratch footprint of every kernel in libcfmm_amd.so (hipcc -Rpass-analysis=kernel-resource-usage on
csrc/sweep_kernels.hip, cross-compiled for gfx950: no GPU needed) and the instruction mix of the ProductTwoCoin tile
loops (fast arithmetic vs the compiler's division / square-root sequences).
usage: python scripts/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cfmmrouter.jl_amd", "csrc", "sweep_kernels.hip")
tmp = tempfile.mkdtemp()
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-save-temps",
       "-Rpass-analysis=kernel-resource-usage", "-c", SRC, "-o", os.path.join(tmp, "k.o")]
r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
print("# hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Rpass-analysis=kernel-resource-usage csrc/sweep_kernels.hip")
print(f"# {len(blocks)} kernels in the code object")
print(f"{'VGPR':>5} {'SGPR':>5} {'scratch':>8} {'LDS(static)':>12} {'waves/SIMD':>11}  kernel")
rows = []
for b in blocks:
    name = b.split("\n")[0].split()[0]
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("cfmm::", "").replace("void ", "")
    dem = re.sub(r"\(.*\)$", "", dem)
    rows.append((dem, g("VGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")))
for dem, v, s_, sc, lds, occ in sorted(rows):
    print(f"{v:5d} {s_:5d} {sc:8d} {lds:12d} {occ:11d}  {dem}")

# instruction classes of the two fused ProductTwoCoin kernels' code (whole kernel: staging, BOTH tile loops -- the one on the
# fast arithmetic and the one on the compiler's division / square-root sequences -- and the epilogue)
asm = open(os.path.join(tmp, "sweep_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
print("\n# static instruction counts (whole kernel code)")
for sym, label in (("_ZN4cfmm12sweep_kernelINS_10ProductOpsELb0ELi1024ELb0EEEvT_NS_9SweepArgsE", "sweep_kernel<ProductOps, false, 1024, false>"),
                   ("_ZN4cfmm12sweep_kernelINS_13GeoMeanLogOpsELb0ELi1024ELb0EEEvT_NS_9SweepArgsE", "sweep_kernel<GeoMeanLogOps, false, 1024, false>"),
                   ("_ZN4cfmm12sweep_kernelINS_8UniV3OpsELb0ELi1024ELb0EEEvT_NS_9SweepArgsE", "sweep_kernel<UniV3Ops, false, 1024, false>")):
    m = re.search(r"^" + sym + r":.*?s_endpgm", asm, flags=re.S | re.M)
    if not m:
        continue
    ins = [l.split()[0] for l in m.group(0).split("\n") if l.startswith("\t") and not l.strip().startswith(";") and not l.strip().startswith(".")]
    cnt = lambda pat: sum(1 for i in ins if re.match(pat, i))
    print(f"  {label}: {len(ins)} instructions; f64 VALU {cnt(r'v_.*_f64')}, of them v_div_scale/fmas/fixup {cnt(r'v_div_')} and "
          f"v_rcp/v_rsq/v_sqrt {cnt(r'v_(rcp|rsq|sqrt)_f64')}; v_cndmask {cnt(r'v_cndmask')}, ds_* {cnt(r'ds_')}, global_* {cnt(r'global_')}, s_* {cnt(r's_')}")
